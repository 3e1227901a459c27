"""Worker for the world_size-2 tests of the N>1 path (block sharding of independent proofs + one gather).

  python -m torch.distributed.run --nproc-per-node 2 tests/_dist_worker.py N          CPU, gloo: sharding and
        gather logic with a stand-in for the prover (libp25 has no CPU path; tests/test_dist_gloo.py)
  ... tests/_dist_worker.py N --real      GPU box: every rank runs the REAL prover on GPU 0 for its shard,
        gloo carries the gather; rank 0 checks all N proofs with the oracle (tests/test_gpu_baseline_configs.py)
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def fake_prove(inputs, words):
    # deterministic "proof": words derived from the input row, so order/completeness is checkable
    return (inputs[:, :1] * 1000 + torch.arange(words, dtype=torch.int64)[None, :])


def main_standin(n_total):
    words = 17
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ge.load_package()
    from plonky25_amd import dist as pd
    start, stop = pd.shard_range(n_total, rank, world)
    all_inputs = torch.arange(n_total, dtype=torch.int64)[:, None].repeat(1, 5)
    local = fake_prove(all_inputs[start:stop], words)
    status = torch.zeros(stop - start, dtype=torch.int32)
    if rank == 1 and stop > start:
        status[0] = 4
    proofs, st = pd.gather_proofs(local, status, n_total)
    t = pd.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == float(world)
    # the bench's form: receive buffers allocated once, used for several steps (results must not leak between steps)
    g = pd.ProofGatherer(n_total, words, torch.device("cpu"))
    for step in range(3):
        blocks, sts = g.gather(local + step, status)
        if rank == 0:
            assert torch.equal(torch.cat(blocks), fake_prove(all_inputs, words) + step)
            assert [b.shape[0] for b in blocks] == pd.shard_sizes(n_total, world)
    # pipelined steps (bench.py): two gathers in flight, one staging slot each
    g2 = pd.ProofGatherer(n_total, words, torch.device("cpu"), slots=2)
    for step in range(4):
        blocks, sts = g2.gather(local + 10 * step, status, slot=step & 1)
        if rank == 0:
            assert torch.equal(torch.cat(blocks), fake_prove(all_inputs, words) + 10 * step)
    # inputs generated on rank 0 only and broadcast (bench.py's plonky3 proof variants)
    src = np.arange(12, dtype=np.uint64).reshape(3, 4) * np.uint64(0x1000000000000001) if rank == 0 else None
    got = pd.broadcast_int64(src, (3, 4), torch.device("cpu"))
    assert got.dtype == np.uint64 and (got == np.arange(12, dtype=np.uint64).reshape(3, 4) * np.uint64(0x1000000000000001)).all()
    if rank == 0:
        expect = fake_prove(all_inputs, words)
        assert proofs.shape == (n_total, words) and torch.equal(proofs, expect)
        s1 = pd.shard_range(n_total, 1, world)[0]
        exp_st = torch.zeros(n_total, dtype=torch.int32)
        if n_total > s1:
            exp_st[s1] = 4
        assert torch.equal(st, exp_st)
        print("DIST_OK", n_total)
    dist.barrier()
    dist.destroy_process_group()


def main_real(n_total):
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    p25 = ge.load_package()
    from plonky25_amd import dist as pd
    p25.device_init(0)                       # both ranks share the box's one GPU
    with open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")) as f:
        base, cfg = p25.p3_proof_from_json(f.read())
    # the same global batch on every rank (deterministic), each rank proves its own block
    variants = [base] + [p25.p3_prove_fibonacci(6, 100, 16, pow_start=v << 24)[0] for v in (1, 2)]
    batch = np.stack([variants[i % 3] for i in range(n_total)])
    bad = n_total - 1                        # the last proof of the last shard carries a corrupted inner proof
    batch[bad, 9000] = (int(batch[bad, 9000]) + 1) % 0xFFFFFFFF00000001
    seeds = np.arange(n_total, dtype=np.uint64) + np.uint64(1000)
    circuit = p25.Circuit.build_p3_verifier(cfg)
    start, stop = pd.shard_range(n_total, rank, world)
    proofs, st = circuit.prove(batch[start:stop], seeds=seeds[start:stop])
    gp, gs = pd.gather_proofs(torch.from_numpy(proofs.view(np.int64)), torch.from_numpy(st), n_total)
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle_binding import Oracle
        oc = Oracle().load_circuit(circuit.to_blob())
        dg, cap = circuit.digest()
        all_p = gp.numpy().view(np.uint64)
        assert gs.tolist() == [0] * (n_total - 1) + [4], gs.tolist()
        for i in range(n_total - 1):
            code, msg = oc.verify(all_p[i], dg, cap)
            assert code == 0, (i, msg)
        # byte equality across the shard boundary: last proof of rank 0's block, first of rank 1's
        s1 = pd.shard_range(n_total, 1, world)[0]
        idx = [s1 - 1, s1]
        po, sto, _per, _wall = oc.prove_many(batch[idx], seeds[idx], threads=2)
        assert (sto == 0).all()
        for k, i in enumerate(idx):
            assert (all_p[i] == po[k]).all(), f"gathered proof {i} differs from the oracle's"
        print("DIST_REAL_OK", n_total)
    dist.barrier()
    dist.destroy_process_group()


def fake_hash(words):
    """A 4-word stand-in for hash_no_pad (order-sensitive, length-sensitive)."""
    w = np.asarray(words, dtype=np.uint64)
    acc = np.array([len(w), 1, 2, 3], dtype=np.uint64)
    for i, v in enumerate(w):
        acc = acc * np.uint64(6364136223846793005) + np.uint64(v) + np.uint64(i * 4 + 1) * np.array([1, 3, 5, 7], dtype=np.uint64)
        acc = np.roll(acc, 1)
    return acc


class StandInCircuit:
    """Has the surface plonky25_amd.aggregate touches.  A "proof" of an aggregator holds fake_hash over its children's
    identifiers as its 4 public inputs (what the real aggregation circuit registers); a leaf has a 64-word cap."""
    WORDS = 80

    class Info:
        degree_bits, num_rows_used = 4, 9

    def __init__(self, child=None, k=0):
        self.child, self.k, self.info = child, k, self.Info()
        self.n_pi = 4 if child is not None else 0

    def build_aggregator(self, k):
        return StandInCircuit(self, k)

    def digest(self):
        return None

    def close(self):
        pass

    def public_inputs(self, proof):
        return proof[-4:].copy()

    fail_k = 0     # an aggregator of exactly this many children fails to prove (the rank-0-only failure test)

    def prove(self, groups, seeds=None):
        from plonky25_amd import aggregate as ag
        if self.k and self.k == StandInCircuit.fail_k:
            raise RuntimeError("stand-in: this aggregate cannot be proved")
        groups = np.asarray(groups, dtype=np.uint64).reshape(-1, self.k * self.WORDS)
        out = np.zeros((groups.shape[0], self.WORDS), dtype=np.uint64)
        for g in range(groups.shape[0]):
            kids = groups[g].reshape(self.k, self.WORDS)
            ids = np.concatenate([fake_hash(ag.leaf_identifier_words(c, self.child.n_pi)) if not self.child.n_pi
                                  else ag.leaf_identifier_words(c, 4) for c in kids])
            out[g, :8] = 777 + g
            out[g, -4:] = fake_hash(ids)
        return out, np.zeros(groups.shape[0], dtype=np.int32)


def main_aggregate(n_per_rank, arity):
    """The sharded aggregation of bench.py on stand-in circuits: shard trees, ONE root per rank gathered, cross-rank
    aggregate on rank 0, and the commitment recomputed from all ranks' leaves."""
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ge.load_package()
    from plonky25_amd import aggregate as ag
    rng = np.random.default_rng(1234)                 # the same global batch on every rank
    all_leaves = rng.integers(0, 1 << 62, size=(world * n_per_rank, StandInCircuit.WORDS), dtype=np.uint64)
    mine = all_leaves[rank * n_per_rank:(rank + 1) * n_per_rank]
    if "--fail-cross" in sys.argv:
        # rank 0's cross-rank aggregate fails (the only aggregator with `world` children: the shard trees' arities are
        # all different from it in this case): EVERY rank must see the error, and every rank must still be able to enter
        # the next collective together (a rank-0-only error used to leave the other ranks in bench.py's
        # all_reduce while rank 0 skipped it)
        assert world not in ag.level_plan(n_per_rank, arity)
        StandInCircuit.fail_k = world
        st = ag.fold_sharded(StandInCircuit(), mine, arity, torch.device("cpu"), True)
        assert st["error"], st
        assert ("cannot be proved" in st["error"]) == (rank == 0)
        t = torch.tensor([rank + 1], dtype=torch.int32)
        dist.all_reduce(t)                                # all ranks arrive here: no hang, no mismatch
        assert int(t.item()) == world * (world + 1) // 2
        if rank == 0:
            print("DIST_AGG_FAIL_AGREED", n_per_rank, arity)
        dist.barrier()
        dist.destroy_process_group()
        return
    st = ag.fold_sharded(StandInCircuit(), mine, arity, torch.device("cpu"), True)
    assert st["error"] is None and st["ranks"] == world and st["leaves_per_rank"] == n_per_rank
    assert [l["arity"] for l in st["fold"]["levels"]] == ag.level_plan(n_per_rank, arity)
    if rank == 0:
        fin = st["final"]
        assert [l["arity"] for l in fin["levels"]] == [world]
        got = [int(v) for v in fin["top"].public_inputs(fin["root"])]
        want = ag.expected_commitment(list(all_leaves[:, :ag.CAP_WORDS]), arity, fake_hash, n_shards=world)
        assert (st["caps"] == all_leaves[:, :ag.CAP_WORDS]).all()
        assert got == want, (got, want)
        # a leaf moved across the shard boundary changes the commitment (the order is the global proof order)
        swapped = all_leaves[:, :ag.CAP_WORDS].copy()
        swapped[[n_per_rank - 1, n_per_rank]] = swapped[[n_per_rank, n_per_rank - 1]]
        assert ag.expected_commitment(list(swapped), arity, fake_hash, n_shards=world) != want
        print("DIST_AGG_OK", n_per_rank, arity)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    n = int(sys.argv[1])
    if "--agg" in sys.argv:
        main_aggregate(n, int(sys.argv[sys.argv.index("--agg") + 1]))
    elif "--real" in sys.argv:
        main_real(n)
    else:
        main_standin(n)
