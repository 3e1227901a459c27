#!/usr/bin/env python3
"""Rows of the aggregation circuit against its number of children (what 2^16 rows hold): profiles/r04_arity.txt."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
inputs, cfg = p25.p3_proof_from_json(open("tests/golden/proof_fibonacci.json").read())
leaf = p25.Circuit.build_p3_verifier(cfg)
print("leaf", int(leaf.info.degree_bits), int(leaf.info.num_rows_used))
for k in (8, 12, 13, 14):
    t=time.time(); a = leaf.build_aggregator(k); i=a.info
    print(k, int(i.degree_bits), int(i.num_rows_used), int(i.num_inputs), round(time.time()-t,2), flush=True)
    if k==13:
        for k2 in (10, 13, 2):
            t=time.time(); b=a.build_aggregator(k2); j=b.info
            print("  L2", k2, int(j.degree_bits), int(j.num_rows_used), round(time.time()-t,2), flush=True)
