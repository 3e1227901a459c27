"""Worker for tests/test_dist_gloo.py: world_size-2 gloo run of the sharding + gather path with a
stand-in prover (the real one needs a GPU; the sharding / gather logic is identical)."""
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


def main():
    n_total, words = int(sys.argv[1]), 17
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = ge.load_package()
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


if __name__ == "__main__":
    main()
