"""CPU: bench.py fails fast -- one JSON error line, non-zero exit -- when fewer GPUs are visible than --gpus asks for."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_fails_fast_without_enough_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 2
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and "64" in d["error"] and d["unit"] == "proofs/s"
