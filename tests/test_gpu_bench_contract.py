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
                        "--cpu-cores", "4", "--aggregate", "4", "--aggregate-arity", "2", "--extra-configs", "none"],
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
    # the roof that binds is VALU issue; achieved / peak / frac are the HBM view the contract prescribes
    assert rf["bound"] == "valu" and rf["contract_view"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "port-tuned", "reference") and cb["value"] > 0 and cb["cores"] == 1
    if cb["kind"] == "port-tuned":      # the AVX-512 leg: validated against the scalar oracle's proof, which stays beside it
        assert cb["bytes_equal_to_the_scalar_oracle_proof"] is True
        assert cb["untuned"]["kind"] == "port" and 0 < cb["untuned"]["value"] < cb["value"]
        assert cb["untuned"]["gpu_proof_bit_exact_vs_this_cpu_proof"] is True
    assert cb["all_cores"]["cores"] == 4 and cb["all_cores"]["value"] > 0
    # the roofline line is the kernel with the GPU to itself; the in-flight figure is reported beside it
    assert rf["launches"] == 8 and rf["avg_launch_ms"] > 0 and rf["avg_launch_ms_timed_region"] >= 0.9 * rf["avg_launch_ms"]
    assert len(d["config"]["oracle_verified_indices"]) == 8
    # the batch folded to one root proof by the recursive verifier circuits, root accepted by the oracle's verifier
    ag = d["aggregation"]
    assert ag["leaves"] == 4 and [l["arity"] for l in ag["levels"]] == [2, 2] and ag["oracle_verifier_accepts_root"] is True
    assert len(ag["root_public_inputs"]) == 4 and ag["root_public_inputs_commit_to_the_leaves"] is True
    assert ag["leaf_equivalent_proofs_per_s_including_aggregation"] > 0
    assert ag["level1_throughput"]["aggregate_proofs_per_s"] > 0          # the level-1 aggregation circuit by itself
    # ... and the same tree resident on the device, pipelined under the leaves (16 leaves, arity 2: four levels)
    pl = ag["pipelined"]
    assert pl["leaf_equivalent_proofs_per_s"] > 0 and pl["all_statuses_ok"] and pl["oracle_verifier_accepts_root"]
    assert pl["root_public_inputs_commit_to_the_leaves"] is True and [l["proofs"] for l in pl["levels"]] == [8, 4, 2, 1]
    # VALU view: priced with the clock measured in the run, and only from a PMC pass of these very kernel sources
    v = rf["valu"]
    assert 1.0e9 < v["shader_clock_hz"] < 3.5e9 and ("stale" in v or v["csrc_sha"])
    assert d["cpu_baseline"]["gpu_proof_bit_exact_vs_this_cpu_proof"] is True


def test_bench_multi_rank_path_bare_launch():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself (before torch
    or HIP is touched).  Test mode: both ranks share GPU 0 and the collectives run on host copies (gloo) -- the rank
    bookkeeping, barriers, max-over-ranks timing, gather and JSON assembly are the code the 8-GPU run executes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "2",
                        "--warmup", "1", "--dist-backend", "gloo", "--aggregate", "4", "--aggregate-arity", "2"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["all_statuses_ok"]
    assert d["config"]["oracle_verifier_accepts"] and "cpu_baseline" not in d
    assert len(d["per_rank"]) == 2 and all(p["proofs_per_s"] > 0 for p in d["per_rank"])
    assert abs(d["value"] - 2 * 8 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the sharded aggregation: each rank folds 4 of its proofs to one root (two 2-to-1 levels), the two roots are
    # gathered, rank 0 proves the 2-to-1 cross-rank aggregate; the root commits to all 8 leaves of both ranks
    ag = d["aggregation"]
    assert ag["leaves"] == 8 and ag["ranks"] == 2 and [l["arity"] for l in ag["levels"]] == [2, 2, 2]
    assert ag["levels"][-1]["level"] == "cross-rank"
    assert ag["root_public_inputs_commit_to_the_leaves"] is True and ag["oracle_verifier_accepts_root"] is True


def test_bench_strong_scaling_mode_two_ranks():
    """--total T (BASELINE config 4 literally: T proofs per step split across the ranks), uneven shards (9 = 5 + 4)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total", "9", "--steps", "1",
                        "--warmup", "1", "--dist-backend", "gloo", "--verify", "9"], capture_output=True, text=True,
                       timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["proofs_per_step_total"] == 9
    assert [p["proofs_per_step"] for p in d["per_rank"]] == [5, 4]
    assert d["config"]["all_statuses_ok"] and d["config"]["gathered_complete_and_ok"] and d["config"]["oracle_verifier_accepts"]
    assert d["config"]["oracle_verified_indices"] == list(range(9))      # every gathered proof, both ranks' blocks
    assert abs(d["value"] - 9 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_four_ranks_uneven_shards_and_the_four_to_one_cross_rank_aggregate():
    """The 4-GPU shape on the one-GPU box (test mode: four ranks on GPU 0, gloo for the collectives): --total 11 splits
    3 + 3 + 3 + 2, every rank folds two of its proofs to a shard root, the FOUR roots are gathered and rank 0 proves the
    4-to-1 cross-rank aggregate (a circuit no two-rank run ever builds); the pipelined trees differ per rank (3 leaves
    against 2) around the same collectives.  Also: what rank 0 does alone after the timed region is reported and far
    below the process group's timeout (the other ranks wait in the final barrier for that long)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--total", "11", "--steps", "2",
                        "--warmup", "1", "--dist-backend", "gloo", "--verify", "11", "--aggregate", "2",
                        "--aggregate-arity", "2"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["proofs_per_step_total"] == 11
    assert [p["proofs_per_step"] for p in d["per_rank"]] == [3, 3, 3, 2]
    assert d["config"]["all_statuses_ok"] and d["config"]["gathered_complete_and_ok"] and d["config"]["oracle_verifier_accepts"]
    assert d["config"]["oracle_verified_indices"] == list(range(11))
    ag = d["aggregation"]
    assert ag["leaves"] == 8 and ag["ranks"] == 4 and [l["arity"] for l in ag["levels"]] == [2, 4]
    assert ag["levels"][-1]["level"] == "cross-rank"
    assert ag["root_public_inputs_commit_to_the_leaves"] is True and ag["oracle_verifier_accepts_root"] is True
    pl = ag["pipelined"]
    assert "error" not in pl, pl
    assert pl["all_statuses_ok"] and pl["oracle_verifier_accepts_root"] and pl["root_public_inputs_commit_to_the_leaves"] is True
    assert 0 < d["rank0_post_region_s"] < 0.25 * d["collective_timeout_s"]
