#!/usr/bin/env python3
"""Throughput of the 8-to-1 aggregation circuit BY ITSELF (batches of aggregate proofs, 16 in flight, steps enqueued back
to back) next to the leaf circuit's, on one box: is an aggregate proof dearer than a leaf proof as a kernel mix, or only
inside the pipelined tree?   usage: agg_throughput.py [aggregates per step = 64] [steps = 3] [--only leaf|agg] [--streams S] [--close-leaf]
(--only: one circuit's timed loop alone, the form to put under rocprofv3 --kernel-trace --stats)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
dev = torch.device("cuda", 0)
only = None
if "--only" in sys.argv:
    i = sys.argv.index("--only"); only = sys.argv[i + 1]; del sys.argv[i:i + 2]
streams = None
if "--streams" in sys.argv:
    i = sys.argv.index("--streams"); streams = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inputs, cfg = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
leaf = p25.Circuit.build_p3_verifier(cfg)
lp, st = leaf.prove(np.stack([inputs] * 8), seeds=list(range(8)))
assert (st == 0).all()


def rate(circ, rows, count):
    pw = int(circ.info.proof_words)
    d_in = torch.from_numpy(np.stack([rows] * count).view(np.int64)).to(dev)
    d_seeds = torch.arange(count, dtype=torch.int64, device=dev)
    d_p = torch.zeros((count, pw), dtype=torch.int64, device=dev)
    d_s = torch.zeros((steps + 1, count), dtype=torch.int32, device=dev)
    for k in range(steps + 1):
        if k == 1:
            circ.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
        circ.prove_dev(d_in.data_ptr(), count, d_seeds.data_ptr(), d_p.data_ptr(), pw, d_s[k].data_ptr())
    circ.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int((d_s != 0).sum().item()) == 0
    return count * steps / dt


agg = leaf.build_aggregator(8)
if streams:
    leaf.set_streams(streams); agg.set_streams(streams)
out = {"streams": streams or "default", "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
if "--close-leaf" in sys.argv:      # the leaf circuit's streams gone before the aggregator runs (hardware-queue sharing)
    assert only == "agg"
    leaf.close(); out["leaf_closed"] = True
if only != "agg":
    out["leaf_proofs_per_s"] = round(rate(leaf, inputs, n if only else 4 * n), 2)
    out["ms_per_leaf_proof"] = round(1e3 / out["leaf_proofs_per_s"], 3)
if only != "leaf":
    out["aggregate_proofs_per_s"] = round(rate(agg, np.concatenate([lp[i] for i in range(8)]), n), 2)
    out["ms_per_aggregate_proof"] = round(1e3 / out["aggregate_proofs_per_s"], 3)
print(json.dumps(out))
