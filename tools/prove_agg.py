#!/usr/bin/env python3
"""Prove a batch of 8-to-1 aggregation proofs (level 1 of the tree: each verifies 8 fib-64 leaf proofs) -- the target
for rocprofv3 kernel traces and PMC passes of the AGGREGATOR's kernels in their throughput forms:
   rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace -- python3 tools/prove_agg.py 8     (then tools/pmc_summary.py dir out.json 8)
usage: prove_agg.py [aggregate proofs in the timed call = 8] [arity = 8]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inputs, cfg = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
c = p25.Circuit.build_p3_verifier(cfg)
leaves, st = c.prove(np.stack([inputs] * (n * k)), seeds=list(range(n * k)))
assert (st == 0).all()
agg = c.build_aggregator(k)
groups = np.stack([np.concatenate([leaves[k * i + j] for j in range(k)]) for i in range(n)])
agg.prove(groups, seeds=list(range(n)))          # warm-up: contexts, tables
proofs, st = agg.prove(groups, seeds=list(range(n)))   # the timed call (pmc_summary.py takes the last k_witgen_set_inputs on)
print(st.tolist(), int(agg.info.degree_bits), agg.gate_counts())
