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


if __name__ == "__main__":
    n = int(sys.argv[1])
    if "--real" in sys.argv:
        main_real(n)
    else:
        main_standin(n)
