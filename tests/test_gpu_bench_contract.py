"""GPU: bench.py prints ONE JSON line with the fields the driver's contract names (small batch)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "16", "--steps", "1", "--warmup", "1",
                        "--cpu-cores", "4"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "proofs/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "workload" in d["config"]
    assert d["config"]["all_statuses_ok"] and d["config"]["oracle_verifier_accepts"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] == 1
    assert cb["all_cores"]["cores"] == 4 and cb["all_cores"]["value"] > 0
    # the roofline line is the kernel with the GPU to itself; the in-flight figure is reported beside it
    assert rf["launches"] == 8 and rf["avg_launch_ms"] > 0 and rf["avg_launch_ms_timed_region"] >= 0.9 * rf["avg_launch_ms"]
    assert len(d["config"]["oracle_verified_indices"]) == 8


def test_bench_multi_rank_path_bare_launch():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself (before torch
    or HIP is touched).  Test mode: both ranks share GPU 0 and the collectives run on host copies (gloo) -- the rank
    bookkeeping, barriers, max-over-ranks timing, gather and JSON assembly are the code the 8-GPU run executes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "1",
                        "--warmup", "1", "--dist-backend", "gloo"], capture_output=True, text=True, timeout=1500,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["all_statuses_ok"]
    assert d["config"]["oracle_verifier_accepts"] and "cpu_baseline" not in d
    assert len(d["per_rank"]) == 2 and all(p["proofs_per_s"] > 0 for p in d["per_rank"])
    assert abs(d["value"] - 2 * 8 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
