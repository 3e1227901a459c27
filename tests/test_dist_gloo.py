"""CPU: the N>1 path (block sharding of independent proofs + final gather) on gloo, world_size 2."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_shard_range(p25):
    from plonky25_amd import dist as pd
    for n, w in ((2048, 8), (256, 2), (7, 3), (1, 2), (0, 4)):
        ranges = [pd.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_total", [6, 7])
def test_gather_world2_gloo(n_total):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29531 + n_total),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_total)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"DIST_OK {n_total}" in out.stdout


@pytest.mark.parametrize("n_per_rank,arity", [(8, 8), (16, 4), (1, 8), (20, 13), (7, 3)])
def test_sharded_aggregation_world2_gloo(n_per_rank, arity):
    """Every rank folds its own shard, ONE root per rank is gathered, rank 0 proves the cross-rank aggregate; the final
    public inputs equal the hash tree over ALL ranks' leaves (stand-in circuits: libp25 has no CPU path; the real
    prover runs the same code in tests/test_gpu_bench_contract.py).  (20, 13) and (7, 3): arities that do not divide
    the shard -- the last group of a level is right-aligned and overlaps its neighbour (aggregate.group_bounds)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29551 + n_per_rank + arity),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_per_rank), "--agg", str(arity)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"DIST_AGG_OK {n_per_rank} {arity}" in out.stdout
