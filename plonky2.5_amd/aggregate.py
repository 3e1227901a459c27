"""The batch folded to ONE proof -- north_star's "final aggregation step" taken literally -- sharded like the batch.

Every level is an aggregation circuit (`p25_circuit_build_aggregator`: upstream `builder.verify_proof` for each of its k
children + four registered public inputs = hash_no_pad over the children's identifiers) proved as a plain batch on the
GPU.  A child's identifier is its own public inputs if it has any (an aggregate further down) and
hash_no_pad(its wires cap) otherwise (a leaf proof of the plonky3-verifier circuit, which registers none,
/root/reference/src/p3/mod.rs:264), so the root exposes the root of a Poseidon tree over the whole batch.

With N ranks every rank folds ITS OWN shard down to one proof on its own GPU (`fold`), the N shard roots -- not the
N x 256 leaves -- cross xGMI (`fold_sharded`: one RCCL gather of N proofs), and rank 0 proves one N-to-1 aggregate on top
(`fold_roots`).  `DeviceTree` is the same tree kept on the device and pipelined under the next step's leaf proving.  `expected_commitment` recomputes what the final root's public inputs must be from the leaves alone; the
hash is passed in (tests and bench.py hand it the oracle's, as the checker)."""
import time

import numpy as np

__all__ = ["fold", "fold_roots", "fold_sharded", "DeviceTree", "expected_commitment", "leaf_identifier_words", "largest_pow2"]

CAP_WORDS = 64   # wires cap = the first 2^cap_height x 4 words of a flat proof (cap_height 4, include/p25.h proof layout)


def largest_pow2(n):
    while n & (n - 1):
        n &= n - 1
    return n


def leaf_identifier_words(proof, n_public_inputs):
    """The words an aggregator hashes for this child: its public inputs, or its wires cap when it has none."""
    return proof[-n_public_inputs:] if n_public_inputs else proof[:CAP_WORDS]


def level_plan(n, arity):
    """Group sizes per level for n children folded `arity` at a time down to one (n a power of two, arity too)."""
    plan = []
    while n > 1:
        k = min(arity, n)
        plan.append(k)
        n //= k
    return plan


def fold(circuit, leaves, arity=8, warm=True, in_flight=16):
    """Folds `leaves` (flat proofs of `circuit`, a power of two of them) into one root proof.  Returns a dict with the
    root proof, the circuit it belongs to (`top`; the caller closes `owned` when done), per-level records and the time
    spent proving (`tree_s`) and building circuits (`build_s`, once per shape).  One leaf: the leaf is the root."""
    n = len(leaves)
    if n < 1 or n & (n - 1):
        raise ValueError("fold needs a power-of-two number of leaves")
    level, circ, levels, owned, tree_s, build_s = list(leaves), circuit, [], [], 0.0, 0.0
    for k in level_plan(n, arity):
        t = time.perf_counter()
        nxt = circ.build_aggregator(k)
        nxt.digest()
        bs = time.perf_counter() - t
        build_s += bs
        owned.append(nxt)
        groups = np.stack([np.concatenate(level[k * i:k * (i + 1)]) for i in range(len(level) // k)])
        if warm:   # this circuit's contexts and tables: once per shape, like the build
            w = min(in_flight, len(groups))
            nxt.prove(groups[:w], seeds=list(range(w)))
        t = time.perf_counter()
        out, st = nxt.prove(groups, seeds=np.arange(len(groups), dtype=np.uint64))
        dt = time.perf_counter() - t
        if not (st == 0).all():
            raise RuntimeError(f"aggregation level {len(levels) + 1}: statuses {st.tolist()}")
        tree_s += dt
        level = [out[i] for i in range(out.shape[0])]
        levels.append({"level": len(levels) + 1, "arity": k, "circuit_rows_log2": int(nxt.info.degree_bits),
                       "rows_used": int(nxt.info.num_rows_used), "proofs": len(level), "prove_s": round(dt, 4),
                       "ms_per_proof": round(dt / len(level) * 1e3, 3), "circuit_build_s": round(bs, 2)})
        circ = nxt
    return {"root": level[0], "top": circ, "owned": owned, "levels": levels, "tree_s": tree_s, "build_s": build_s}


def fold_roots(top, roots, warm=True):
    """Rank 0's last step: one aggregate over the N shard roots (all proofs of `top`, the shard trees' top circuit --
    identical on every rank because every shard has the same shape).  N = 1: nothing to do."""
    n = len(roots)
    if n == 1:
        return {"root": roots[0], "top": top, "owned": [], "levels": [], "tree_s": 0.0, "build_s": 0.0}
    t = time.perf_counter()
    fin = top.build_aggregator(n)
    fin.digest()
    bs = time.perf_counter() - t
    group = np.concatenate(roots)[None, :]
    if warm:
        fin.prove(group, seeds=[0])
    t = time.perf_counter()
    out, st = fin.prove(group, seeds=[0])
    dt = time.perf_counter() - t
    if int(st[0]) != 0:
        raise RuntimeError(f"cross-rank aggregation: status {int(st[0])}")
    rec = {"level": "cross-rank", "arity": n, "circuit_rows_log2": int(fin.info.degree_bits),
           "rows_used": int(fin.info.num_rows_used), "proofs": 1, "prove_s": round(dt, 4),
           "ms_per_proof": round(dt * 1e3, 3), "circuit_build_s": round(bs, 2)}
    return {"root": out[0], "top": fin, "owned": [fin], "levels": [rec], "tree_s": dt, "build_s": bs}


def expected_commitment(leaf_caps, arity, hash_no_pad, n_shards=1):
    """What the final root's four public inputs must be, from the leaves alone: `leaf_caps` = the wires caps of ALL
    leaves in global order (shard after shard), each shard folded `arity` at a time, the shard roots folded once more
    when n_shards > 1 -- the tree of the aggregation's own shape."""
    per = len(leaf_caps) // n_shards
    roots = []
    for s in range(n_shards):
        ids = [hash_no_pad(np.asarray(c, dtype=np.uint64)) for c in leaf_caps[s * per:(s + 1) * per]]
        for k in level_plan(per, arity):
            ids = [hash_no_pad(np.concatenate(ids[k * i:k * (i + 1)])) for i in range(len(ids) // k)]
        roots.append(ids[0])
    if n_shards == 1:
        return [int(v) for v in roots[0]]
    return [int(v) for v in hash_no_pad(np.concatenate(roots))]


def fold_sharded(circuit, local_leaves, arity, cdev, distributed):
    """All ranks call this with the same number of local leaves (flat proofs of `circuit`, [n, words] uint64).  Every
    rank folds its shard on its own GPU; the N shard roots and -- for the check -- the leaves' wires caps are gathered
    onto rank 0, which proves the cross-rank aggregate.  Collectives only where every rank reaches them: a local
    failure is agreed on first.  Returns a state dict: on every rank `fold` (its shard tree), `error`, `tree_s_max`,
    `roots_gather_ms`; on rank 0 also `final` (fold_roots) and `caps` (all leaves' caps in global order)."""
    import torch
    import torch.distributed as dist
    from . import dist as pdist
    n = int(local_leaves.shape[0])
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    f, err = None, None
    try:
        f = fold(circuit, [local_leaves[i] for i in range(n)], arity=arity)
    except Exception as e:   # the other ranks must learn of it before anyone enters a collective
        err = str(e)[:300]
    all_fine = err is None
    if distributed:
        fl = torch.tensor([1 if all_fine else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(fl, op=dist.ReduceOp.MIN)
        all_fine = bool(fl.item())
    state = {"error": err or (None if all_fine else "another rank's shard tree failed"), "fold": f,
             "leaves_per_rank": n, "ranks": world}
    if not all_fine:
        return state
    roots, caps, gather_ms, tree_s_max = [f["root"]], np.ascontiguousarray(local_leaves[:, :CAP_WORDS]), 0.0, f["tree_s"]
    if distributed:
        tree_s_max = pdist.max_over_ranks(f["tree_s"], cdev)
        g0 = time.perf_counter()
        rg = pdist.ProofGatherer(world, int(f["root"].size), cdev)      # ONE root proof per rank crosses xGMI
        r_blocks, _ = rg.gather(torch.from_numpy(f["root"].view(np.int64).copy())[None, :].to(cdev),
                                torch.zeros(1, dtype=torch.int32, device=cdev))
        if cdev.type == "cuda":
            torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        cg = pdist.ProofGatherer(world * n, CAP_WORDS, cdev)            # the leaves' wires caps: the checker's input
        c_blocks, _ = cg.gather(torch.from_numpy(caps.view(np.int64).copy()).to(cdev),
                                torch.zeros(n, dtype=torch.int32, device=cdev))
        if rank == 0:
            roots = [b[0].cpu().numpy().view(np.uint64) for b in r_blocks]
            caps = np.concatenate([b.cpu().numpy().view(np.uint64) for b in c_blocks])
    if rank == 0:
        try:
            state["final"] = fold_roots(f["top"], roots)
        except Exception as e:
            state["error"] = str(e)[:300]
    state.update({"caps": caps, "tree_s_max": tree_s_max, "roots_gather_ms": gather_ms})
    return state


class DeviceTree:
    """The aggregation tree of one shard kept RESIDENT ON THE DEVICE and only ever enqueued: an aggregation circuit's
    inputs are its k children's flat proofs back to back, i.e. exactly k consecutive rows of the buffer the level
    below writes its proofs into (proof stride = proof words), so a level is `p25_prove_batch_dev` straight on the
    previous level's output -- no host round trip, no synchronisation.  Ordering between the circuits' streams is
    device-side (`p25_circuit_stream_join` / `p25_circuit_wait_stream`), so the tree of step j runs underneath the leaf
    proving of step j+1 and its latency-bound top (one or two proofs per level) costs no idle machine.

    Buffers are `slots`-deep: step j uses slot j % slots.  A level's proofs of step j are overwritten by step j+slots,
    which first waits (on the device) for the level ABOVE to have finished step j."""

    def __init__(self, circuit, n_leaves, arity, device, slots=2):
        import torch
        if n_leaves < 2 or n_leaves & (n_leaves - 1):
            raise ValueError("DeviceTree needs a power-of-two number of leaves >= 2")
        self.torch, self.dev, self.slots, self.n_leaves, self.arity = torch, device, slots, n_leaves, arity
        self.leaf, self.levels, self.build_s = circuit, [], 0.0
        circ, n = circuit, n_leaves
        for k in level_plan(n_leaves, arity):
            t = time.perf_counter()
            nxt = circ.build_aggregator(k)
            nxt.digest()
            self.build_s += time.perf_counter() - t
            n //= k
            pw = int(nxt.info.proof_words)
            assert int(nxt.info.num_inputs) == k * int(circ.info.proof_words)
            self.levels.append({
                "circ": nxt, "n": n, "k": k, "pw": pw,
                "seeds": torch.arange(n, dtype=torch.int64, device=device),
                "out": [torch.zeros((n, pw), dtype=torch.int64, device=device) for _ in range(slots)],
                "status": [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(slots)],
                # marker[s]: a stream that holds "this level's work up to the latest step of slot s"
                "marker": [torch.cuda.Stream(device=device) for _ in range(slots)],
            })
            circ = nxt
        self.top = circ
        self.hop = [torch.cuda.Stream(device=device) for _ in self.levels]   # carry "the level below is done" upwards
        self.aggregates_per_step = sum(L["n"] for L in self.levels)
        self.steps = 0

    def before_leaves(self):
        """Call before enqueuing the leaf proofs of the next step into its slot: they overwrite the buffer level 1 read
        `slots` steps ago."""
        if self.steps >= self.slots:
            self.leaf.wait_stream(self.levels[0]["marker"][self.steps % self.slots].cuda_stream)

    def enqueue(self, d_leaf_proofs):
        """Enqueue the whole tree over the leaf proofs just enqueued (`d_leaf_proofs`: [n_leaves, leaf proof words]
        int64, contiguous, being written by the leaf circuit's streams).  Returns the slot."""
        s = self.steps % self.slots
        below_c, below_buf = self.leaf, d_leaf_proofs
        for i, L in enumerate(self.levels):
            c = L["circ"]
            below_c.stream_join(self.hop[i].cuda_stream)       # everything the level below has been asked for ...
            c.wait_stream(self.hop[i].cuda_stream)             # ... before this level's witness generation reads it
            if self.steps >= self.slots and i + 1 < len(self.levels):   # the level above still reads out[s] of step - slots
                c.wait_stream(self.levels[i + 1]["marker"][s].cuda_stream)
            c.prove_dev(below_buf.data_ptr(), L["n"], L["seeds"].data_ptr(), L["out"][s].data_ptr(), L["pw"],
                        L["status"][s].data_ptr())
            c.stream_join(L["marker"][s].cuda_stream)
            below_c, below_buf = c, L["out"][s]
        self.steps += 1
        return s

    def sync(self):
        for L in self.levels:
            L["circ"].sync()

    def root(self, slot):
        """(root proof, all statuses zero) of a finished step (host copies)."""
        ok = all(int((L["status"][slot] != 0).sum().item()) == 0 for L in self.levels)
        return self.levels[-1]["out"][slot][0].cpu().numpy().view(np.uint64), ok

    def close(self):
        for L in self.levels:
            L["circ"].close()
