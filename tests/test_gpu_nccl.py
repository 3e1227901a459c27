"""GPU: the RCCL (`nccl`) branch executes for real on the one-GPU box -- a world-size-1 process group.

No 8-GPU node is available to the builder, so before round 4 no `nccl` process group had ever been initialised by this
code.  These tests take every collective of the multi-GPU path through RCCL on device tensors, and run bench.py's
distributed branch (`--force-dist`): pipelined steps with the gather of step k on a side stream underneath step k+1."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


def test_rccl_world1_collectives_on_device_tensors():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_nccl_worker.py")], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=_env())
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "NCCL_WORLD1_OK" in r.stdout


def test_bench_force_dist_takes_the_rccl_branch():
    """bench.py --gpus 1 --force-dist: RCCL init with device_id, per-step device-tensor gather on the side streams
    (5 steps: the buffer-reuse wait of step k >= 2 executes), the float64 reductions, the sharded-aggregation collectives."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--batch", "24",
                        "--steps", "3", "--warmup", "2", "--cpu-baseline", "none", "--extra-configs", "none",
                        "--aggregate", "4", "--aggregate-arity", "2", "--verify", "4"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["all_statuses_ok"]
    assert d["config"]["gathered_complete_and_ok"] and d["config"]["oracle_verifier_accepts"]
    assert "TEST MODE" not in d["config"]["workload"]
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["proofs_per_step"] == 24 and d["per_rank"][0]["gather_ms_per_step"] > 0
    assert abs(d["value"] - 24 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    ag = d["aggregation"]
    assert ag["leaves"] == 4 and ag["ranks"] == 1 and ag["root_public_inputs_commit_to_the_leaves"] is True
    assert ag["oracle_verifier_accepts_root"] is True


def test_bench_dist_native_gathers_through_the_librarys_own_communicator():
    """bench.py --gpus 1 --force-dist --dist-native: the gather of every step through p25_comm_init / p25_gather_proofs
    (librccl called by libp25 on its own side stream, ordered by p25_circuit_mark) instead of torch.distributed; the gathered
    proofs are the ones the oracle's verifier is handed, so a gather that moved nothing fails the run."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-native", "--batch", "24",
                        "--steps", "4", "--warmup", "2", "--cpu-baseline", "none", "--extra-configs", "none",
                        "--aggregate", "0", "--verify", "6"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["all_statuses_ok"]
    assert d["config"]["gathered_complete_and_ok"] and d["config"]["oracle_verifier_accepts"]
    assert "p25_gather_proofs" in d["config"]["workload"]
    assert d["per_rank"][0]["proofs_per_step"] == 24 and d["per_rank"][0]["gather_ms_per_step"] > 0
    assert d["config"]["runtime"]["hw_queues_env"] == 24 and d["config"]["runtime"]["hw_queues_setting_late"] == 0
