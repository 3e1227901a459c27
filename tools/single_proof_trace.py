#!/usr/bin/env python3
"""Summarise the kernel trace of ONE proof alone on the GPU (config 2):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_sp -- python3 tools/prove_one.py 1
    python3 tools/single_proof_trace.py gpurun_out/_sp > profiles/<tag>_single_proof_kernel_trace_summary.txt
Takes the launches of the LAST prove call (from its k_witgen_set_inputs on)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("p25::", "").replace("void ", "")) for r in kt))
last = max(i for i, r in enumerate(rows) if "k_witgen_set_inputs" in r[2])
rows = rows[last:]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
print("rocprofv3 --kernel-trace -- python3 tools/prove_one.py 1   (ONE fib-64 proof alone on the MI355X, latency-oriented kernel forms)")
print(f"launches {len(rows)}   first start -> last end {span:.3f} ms   sum of kernel durations {tot:.3f} ms   idle between kernels {span - tot:.3f} ms")
print(f"{'kernel':34s} {'launches':>8s} {'total ms':>10s} {'average us':>12s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:34]:34s} {c:8d} {t:10.3f} {t / c * 1e3:12.1f}")
