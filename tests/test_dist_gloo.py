"""CPU: the N>1 path (block sharding of independent proofs + final gather) on gloo, world_size 2, 4 and 8."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port


def test_shard_range(p25):
    from plonky25_amd import dist as pd
    for n, w in ((2048, 8), (256, 2), (7, 3), (1, 2), (0, 4)):
        ranges = [pd.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1


def run_ranks(world, *worker_args, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py")] + [str(a) for a in worker_args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("world,n_total", [(4, 11), (8, 2048), (8, 13)])
def test_gather_world4_and_8_gloo(world, n_total):
    """The shapes of the 4- and 8-GPU runs: uneven shards (11 = 3+3+3+2, 13 = 2x5 + 1x3) and BASELINE config 4's 2048 = 8 x 256."""
    assert f"DIST_OK {n_total}" in run_ranks(world, n_total)


@pytest.mark.parametrize("world,n_per_rank,arity", [(4, 5, 3), (8, 3, 2), (8, 20, 13), (4, 1, 8)])
def test_sharded_aggregation_world4_and_8_gloo(world, n_per_rank, arity):
    """N shard trees, N roots gathered, ONE N-to-1 aggregate on rank 0 (N = 4, 8), commitment over all N shards' leaves."""
    assert f"DIST_AGG_OK {n_per_rank} {arity}" in run_ranks(world, n_per_rank, "--agg", arity)


@pytest.mark.parametrize("world", [2, 4])
def test_cross_rank_aggregate_failure_is_agreed_on_by_every_rank(world):
    """rank 0's cross-rank aggregate failing must not leave the ranks on different paths around the
    collectives that follow (bench.py gates its pipelined-tree block, which holds collectives, on the agreed outcome)."""
    assert "DIST_AGG_FAIL_AGREED 9 3" in run_ranks(world, 9, "--agg", 3, "--fail-cross")


@pytest.mark.parametrize("n_total", [6, 7])
def test_gather_world2_gloo(n_total):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
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
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_per_rank), "--agg", str(arity)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"DIST_AGG_OK {n_per_rank} {arity}" in out.stdout
