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

__all__ = ["fold", "fold_roots", "fold_sharded", "release_fold", "DeviceTree", "circuit_throughput", "expected_commitment", "leaf_identifier_words", "level_plan", "group_bounds", "widest_arity"]

MAX_MARKS = 16                    # include/p25.h P25_MAX_MARKS
TREE_MARK_SLOTS = MAX_MARKS - 2   # DeviceTree uses slots 0 .. 13 (levels + 2 of them): 256 leaves two at a time (8 levels) fit
CAP_WORDS = 64   # wires cap = the first 2^cap_height x 4 words of a flat proof (cap_height 4, include/p25.h proof layout)


def widest_arity(circuit, cap=16):
    """The number of children (<= cap) at which an aggregation circuit over proofs of `circuit` costs least per child:
    an aggregation circuit pays for 2^ceil(log2 rows) rows and uses rows(k) = a + b k of them (the verifier of one more
    child is the same rows again), so the best k fills a power of two.  a and b come from two small builds (k = 2, 3),
    the chosen k is built and checked (one smaller if the estimate was a few rows short).  fib-64 verifier proofs: 13
    (62,753 of 2^16 rows; 8 use 38,687, 14 need 2^17) -- profiles/r04_arity.txt."""
    rows = {}
    for k in (2, 3):
        c = circuit.build_aggregator(k)
        try:
            rows[k] = int(c.info.num_rows_used)
        finally:
            c.close()
    r2, r3 = rows[2], rows[3]
    b, a = r3 - r2, r2 - 2 * (r3 - r2)

    def padded(k):
        return 1 << max(1, (a + b * k - 1).bit_length())

    best = min(range(2, cap + 1), key=lambda k: (padded(k) / k, -k))
    while best > 2:
        c = circuit.build_aggregator(best)
        try:
            fits = (1 << int(c.info.degree_bits)) <= padded(best)
        finally:
            c.close()
        if fits:
            break
        best -= 1
    return best


def leaf_identifier_words(proof, n_public_inputs):
    """The words an aggregator hashes for this child: its public inputs, or its wires cap when it has none."""
    return proof[-n_public_inputs:] if n_public_inputs else proof[:CAP_WORDS]


def level_plan(n, arity):
    """Group sizes per level for n children folded at most `arity` at a time down to one: a level of n children is
    ceil(n / arity) groups of k = ceil(n / groups) children each.  Powers of two fold exactly (256 by 8: 8, 8, 4).  When
    groups x k > n -- 256 by 13: 20 groups of 13 = 260 -- the LAST group is the last k children, i.e. it overlaps its
    neighbour and re-verifies up to k - 1 children that group has verified already (`group_bounds`): every level stays
    one batch of one circuit over consecutive rows, and nothing is padded or copied."""
    plan = []
    while n > 1:
        groups = -(-n // arity)
        k = -(-n // groups)
        plan.append(k)
        n = groups
    return plan


def group_bounds(n, k):
    """[(first child, one past the last)] of the ceil(n / k) groups of a level of n children: consecutive, the last
    one right-aligned (it overlaps the one before it when k does not divide n)."""
    groups = -(-n // k)
    return [(k * i, k * (i + 1)) for i in range(groups - 1)] + [(n - k, n)]


def fold(circuit, leaves, arity=8, warm=True, in_flight=16, streams=None, circuits=None):
    """Folds `leaves` (flat proofs of `circuit`, any number of them: `level_plan`) into one root proof.  Returns a dict with the
    root proof, the circuit it belongs to (`top`; the caller closes `owned` when done), per-level records and the time
    spent proving (`tree_s`) and building circuits (`build_s`, once per shape).  One leaf: the leaf is the root.
    `streams`: proofs every level circuit keeps in flight (None = the library's 16; fewer when several processes share one GPU's
    memory).  `circuits`: the level circuits of an earlier fold of the SAME shape (its `owned`), reused instead of rebuilt --
    the shards of one batch all fold through the same circuits; they stay the earlier fold's to close."""
    n = len(leaves)
    if n < 1:
        raise ValueError("fold needs at least one leaf")
    level, circ, levels, owned, tree_s, build_s = list(leaves), circuit, [], [], 0.0, 0.0
    plan = level_plan(n, arity)
    if circuits is not None and len(circuits) != len(plan):
        raise ValueError("fold: `circuits` does not have one circuit per level of this shape")
    for li, k in enumerate(plan):
        t = time.perf_counter()
        if circuits is not None:
            nxt = circuits[li]
            if int(nxt.info.num_inputs) != k * int(circ.info.proof_words):
                raise ValueError(f"fold: circuit {li} of `circuits` does not aggregate {k} proofs of the level below")
        else:
            nxt = circ.build_aggregator(k)
            nxt.digest()
            if streams:
                nxt.set_streams(max(1, int(streams)))
            owned.append(nxt)
        bs = time.perf_counter() - t
        build_s += bs
        n_children = len(level)
        groups = np.stack([np.concatenate(level[a:b]) for a, b in group_bounds(n_children, k)])
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
        levels.append({"level": len(levels) + 1, "arity": k, "children": n_children, "circuit_rows_log2": int(nxt.info.degree_bits),
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
            ids = [hash_no_pad(np.concatenate(ids[a:b])) for a, b in group_bounds(len(ids), k)]
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
    if distributed:
        # rank 0 alone proves the cross-rank aggregate: its outcome is agreed on, so that every rank takes the same
        # path around whatever collectives the caller issues next (a rank-0-only error used to leave the others in them)
        fl = torch.tensor([0 if state["error"] else 1], dtype=torch.int32, device=cdev)
        dist.all_reduce(fl, op=dist.ReduceOp.MIN)
        if not bool(fl.item()) and not state["error"]:
            state["error"] = "rank 0's cross-rank aggregate failed"
    state.update({"caps": caps, "tree_s_max": tree_s_max, "roots_gather_ms": gather_ms})
    return state


def release_fold(state):
    """Keep what the checker needs of a fold_sharded state -- on rank 0 the final root, its circuit's blob, verifier data
    and public inputs -- and close every aggregation circuit the fold built (their per-proof working sets are tens of GB
    that the pipelined tree measured next should not have to share the device with)."""
    fin = state.get("final")
    if fin is not None:
        top, root = fin["top"], fin["root"]
        dg, cap = top.digest()
        state["checker"] = {"root": root, "top_blob": top.to_blob(), "digest": dg, "cap": cap,
                            "root_public_inputs": [int(v) for v in top.public_inputs(root)],
                            "levels": state["fold"]["levels"] + fin["levels"], "final_tree_s": fin["tree_s"],
                            "build_s": state["fold"]["build_s"] + fin["build_s"]}
    owned = list((state.get("fold") or {}).get("owned", [])) + list((fin or {}).get("owned", []))
    for c in owned:
        c.close()
    if state.get("fold"):
        state["fold"]["owned"], state["fold"]["top"] = [], None
    if fin is not None:
        fin["owned"], fin["top"] = [], None
    return state


def circuit_throughput(circ, input_row, device, count=64, steps=2):
    """Proofs/s of `circ` BY ITSELF in its throughput form: `count` proofs of the same inputs per step (distinct filler
    seeds), `steps` steps enqueued back to back after one warm-up step, one synchronisation at the end -- what one proof
    of this circuit costs the machine when enough of them are in flight (an aggregation level in isolation)."""
    import torch
    pw = int(circ.info.proof_words)
    d_in = torch.from_numpy(np.stack([np.asarray(input_row, dtype=np.uint64)] * count).view(np.int64)).to(device)
    d_seeds = torch.arange(count, dtype=torch.int64, device=device)
    d_p = torch.zeros((count, pw), dtype=torch.int64, device=device)
    d_s = torch.zeros((steps + 1, count), dtype=torch.int32, device=device)
    t0 = 0.0
    for k in range(steps + 1):
        if k == 1:
            circ.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        circ.prove_dev(d_in.data_ptr(), count, d_seeds.data_ptr(), d_p.data_ptr(), pw, d_s[k].data_ptr())
    circ.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int((d_s != 0).sum().item()) != 0:
        raise RuntimeError("circuit_throughput: a proof failed")
    return count * steps / dt


class DeviceTree:
    """The aggregation tree of one shard kept RESIDENT ON THE DEVICE and only ever enqueued: an aggregation circuit's
    inputs are its k children's flat proofs back to back, i.e. exactly k consecutive rows of the buffer the level
    below writes its proofs into (proof stride = proof words), so a level is ONE `p25_prove_batch_dev_windows` straight on the
    previous level's output -- no host round trip, no synchronisation.  Any number of leaves and any arity: when k does
    not divide a level's children its last group is the last k rows (`group_bounds`), the right-aligned last window of the same batch.  Ordering between the circuits is device-side
    and event-only (`p25_circuit_mark` / `p25_circuit_wait_mark`).

    The schedule is LAGGED: level l of step i is enqueued one step after level l-1 of step i (level 1 after the leaves
    of step i+1 have been enqueued, level 2 after those of step i+2, ...).  HIP streams are multiplexed onto a few
    hardware queues, and a wait that is not yet satisfied when it reaches the head of its queue holds up whatever was
    enqueued behind it on the streams sharing that queue -- the next step's leaf kernels; issued a step late, the wait
    is long satisfied and costs nothing, and the tree's latency-bound top (one or two proofs per level) hides under a
    full machine.  (Round 4, measured: un-lagged with helper streams 110 leaf-equivalent proofs/s against 138
    un-aggregated -- 12.8 ms of machine per aggregate proof for 7.4 ms of work; more hardware queues: 70.)

    Buffers are `slots` = levels + 2 deep: step i uses slot i % slots; a buffer is rewritten only after the level above
    has marked that it has finished reading it.  `level_streams`: proofs kept in flight by level 1, 2, ... (the last
    entry repeats); None = the library's 16 for every level.  All circuits of a process prove on ONE pool of 16 streams
    (prover.hip: StreamPool), so a level's proofs are interleaved with the leaf proofs on those streams; spread over
    all 16 (None) the tree measured 119.5-120.0 leaf-equivalent proofs/s, concentrated on 4 / 2 / 1 of them 117.1
    (profiles/r04_stream_pool.txt)."""

    def __init__(self, circuit, n_leaves, arity, device, leaf_batch=None, level_streams=None):
        import torch
        if n_leaves < 2:
            raise ValueError("DeviceTree needs at least two leaves")
        self.torch, self.dev, self.n_leaves, self.arity = torch, device, n_leaves, arity
        self.leaf, self.levels, self.build_s = circuit, [], 0.0
        plan = level_plan(n_leaves, arity)
        self.slots = len(plan) + 2
        if self.slots > TREE_MARK_SLOTS:     # the top two mark slots are the caller's (bench.py: the gather of the timed steps)
            raise ValueError(f"tree too deep for the mark slots it may use (0..{TREE_MARK_SLOTS - 1}): {len(plan)} levels")
        self.leaf_batch = leaf_batch or n_leaves
        lpw = int(circuit.info.proof_words)
        self.leaf_buf = [torch.zeros((self.leaf_batch, lpw), dtype=torch.int64, device=device) for _ in range(self.slots)]
        circ, n = circuit, n_leaves
        for k in plan:
            t = time.perf_counter()
            nxt = circ.build_aggregator(k)
            nxt.digest()
            self.build_s += time.perf_counter() - t
            n_children, n = n, -(-n // k)
            if level_streams:
                nxt.set_streams(max(1, min(level_streams[min(len(self.levels), len(level_streams) - 1)], n)))
            pw = int(nxt.info.proof_words)
            assert int(nxt.info.num_inputs) == k * int(circ.info.proof_words)
            self.levels.append({
                "circ": nxt, "n": n, "k": k, "pw": pw, "children": n_children, "cpw": int(circ.info.proof_words),
                "seeds": torch.arange(n, dtype=torch.int64, device=device),
                "out": [torch.zeros((n, pw), dtype=torch.int64, device=device) for _ in range(self.slots)],
                "status": [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(self.slots)],
            })
            circ = nxt
        self.top = circ
        self.aggregates_per_step = sum(L["n"] for L in self.levels)
        self.host_steps = 0      # calls of step()
        self.leaf_steps = 0      # steps whose leaves have been enqueued

    def step(self, prove_leaves=None):
        """One host step: enqueue the leaf proofs of step j (`prove_leaves(buffer, j)` must enqueue them on the leaf
        circuit into `buffer`, [leaf_batch, proof words] int64; the tree folds its first n_leaves rows), then level l of
        step j - l for every level.  `prove_leaves=None`: no new leaves (flushing the levels still owed)."""
        j, S = self.host_steps, self.slots
        circs = [self.leaf] + [L["circ"] for L in self.levels]
        if prove_leaves is not None:
            assert j == self.leaf_steps, "leaf steps must be consecutive"
            s = j % S
            if j >= S:                                   # level 1 has finished reading this slot (step j - S)
                self.leaf.wait_mark(circs[1], s)
            prove_leaves(self.leaf_buf[s], j)
            self.leaf.mark(s)
            self.leaf_steps += 1
        for l in range(1, len(circs)):
            i = j - l
            if i < 0 or i >= self.leaf_steps:
                continue
            s, L, c = i % S, self.levels[l - 1], circs[l]
            c.wait_mark(circs[l - 1], s)                 # the level below has written step i (marked a host step ago)
            if i >= S and l + 1 < len(circs):            # the level above has finished reading out[s] of step i - S
                c.wait_mark(circs[l + 1], s)
            below = self.leaf_buf[s][:self.n_leaves] if l == 1 else self.levels[l - 2]["out"][s]
            # groups 0 .. n-2 are consecutive windows of k rows of `below`, the last one its last k rows (group_bounds): ONE
            # batch either way (p25_prove_batch_dev_windows; a separate batch-of-one call for the overlapping group, as rounds
            # 3-4 issued it, runs that proof alone with the latency-oriented kernel forms)
            n, k, pw = L["n"], L["k"], L["pw"]
            c.prove_dev_windows(below.data_ptr(), k * L["cpw"], (L["children"] - k) * L["cpw"], n, L["seeds"].data_ptr(),
                                L["out"][s].data_ptr(), pw, L["status"][s].data_ptr())
            c.mark(s)
        self.host_steps += 1

    def flush(self):
        """Enqueue the levels still owed to the leaf steps already enqueued."""
        for _ in self.levels:
            self.step(None)

    def sync(self):
        self.leaf.sync()
        for L in self.levels:
            L["circ"].sync()

    def root(self, step):
        """(root proof, every aggregate's status zero) of leaf step `step` (one of the last `slots`), after sync()."""
        s = step % self.slots
        ok = all(int((L["status"][s] != 0).sum().item()) == 0 for L in self.levels)
        return self.levels[-1]["out"][s][0].cpu().numpy().view(np.uint64), ok

    def leaf_proofs(self, step):
        return self.leaf_buf[step % self.slots]

    def close(self):
        for L in self.levels:
            L["circ"].close()
