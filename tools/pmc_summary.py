#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc ... --kernel-trace CSV pair per kernel (second half of the run = the
timed prove call of tools/prove_one.py).  usage: pmc_summary.py <dir with *_counter_collection.csv> [out.json] [proofs in the timed call]
All figures are per proof."""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0])))
kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0])))
# the timed prove call starts at the LAST k_witgen_set_inputs dispatch; counts are divided by its batch size
n_proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
starts = [int(r["Dispatch_Id"]) for r in kt if "k_witgen_set_inputs" in r["Kernel_Name"]]
half = max(starts) - 1 if starts else sorted(int(r["Dispatch_Id"]) for r in kt)[len(kt) // 2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in kt:
    if int(r["Dispatch_Id"]) <= half: continue
    k = r["Kernel_Name"].split("(")[0]
    agg[k]["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k]["calls"] += 1
for r in cc:
    if int(r["Dispatch_Id"]) <= half: continue
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
for v in agg.values():
    for k in list(v):
        v[k] /= n_proofs
rows = sorted(agg.items(), key=lambda kv: -kv[1]["dur_ns"])
names = sorted({c for _, v in rows for c in v if c not in ("dur_ns", "calls")})
print(f"{'kernel':28s} calls  dur_ms " + " ".join(f"{n[:14]:>14s}" for n in names))
for k, v in rows:
    print(f"{k[:28]:28s} {int(v['calls']):5d} {v['dur_ns']/1e6:7.3f} " + " ".join(f"{v[n]/1e6:14.2f}" for n in names))
tot = {n: sum(v[n] for _, v in rows) for n in names}
print("TOTAL (millions):", {n: round(t / 1e6, 1) for n, t in tot.items()}, "dur_ms", round(sum(v["dur_ns"] for _, v in rows) / 1e6, 3))
if len(sys.argv) > 2:
    # _meta ties the counts to the kernel sources they were collected for (bench.py's VALU view checks it)
    import hashlib, subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(ROOT, "plonky2.5_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h", ".inc")) and fn != "capi.hip":   # capi.hip: host code only (the C ABI), no kernel
            h.update(open(os.path.join(csrc, fn), "rb").read())
    out = {k: dict(v) for k, v in rows}
    out["_meta"] = {"csrc_sha": h.hexdigest()[:16], "proofs_in_timed_call": n_proofs,
                    "head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
