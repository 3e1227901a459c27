#!/bin/bash
# Which kernels make an aggregate proof dearer than a leaf proof in the THROUGHPUT form?  The same loop (64 proofs per
# step, 16 in flight, steps back to back) for each circuit alone under rocprofv3 --kernel-trace --stats; the per-kernel
# sums of (overlapping) durations are "stream time": with 16 streams always busy they add up to ~16 x the wall time.
#   tools/agg_vs_leaf.sh <tag>     -> gpurun_out/<tag>_agg_vs_leaf.txt
set -u
TAG=${1:-rXX}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for W in leaf agg; do
  rm -rf $OUT/_prof_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof_$W -- python3 tools/agg_throughput.py 64 3 --only $W > $OUT/_avl_$W.json 2> $OUT/_avl_$W.err
  find $OUT/_prof_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/_avl_$W.csv
  rm -rf $OUT/_prof_$W
done
python3 - $OUT > $OUT/${TAG}_agg_vs_leaf.txt <<'P'
import csv, sys, json, re
out = sys.argv[1]
def load(w):
    d = {}
    for r in csv.DictReader(open(f"{out}/_avl_{w}.csv")):
        name = re.sub(r"\(.*", "", r["Name"]).replace("p25::", "").replace("void ", "").strip()
        d[name] = d.get(name, 0.0) + float(r["TotalDurationNs"]) / 1e6
    return d, open(f"{out}/_avl_{w}.json").read().strip()
L, lj = load("leaf"); A, aj = load("agg")
print("leaf:", lj); print("agg: ", aj)
nl = 4 * 64; na = 4 * 64 + 8   # proofs traced: (steps + 1) x 64; the aggregator run also proves 8 leaves first
print(f"{'kernel':42s} {'leaf ms/proof':>14s} {'agg ms/proof':>14s} {'diff':>8s}   (sums of overlapping durations, 16 streams)")
tl = ta = 0.0
for k in sorted(set(L) | set(A), key=lambda k: -(A.get(k, 0) / na - L.get(k, 0) / nl)):
    l, a = L.get(k, 0) / nl, A.get(k, 0) / na
    tl += l; ta += a
    if max(l, a) > 0.05: print(f"{k[:42]:42s} {l:14.3f} {a:14.3f} {a - l:8.3f}")
print(f"{'total':42s} {tl:14.3f} {ta:14.3f} {ta - tl:8.3f}")
P
cat $OUT/${TAG}_agg_vs_leaf.txt
