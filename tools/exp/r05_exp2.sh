#!/bin/bash
# Round 5, experiment 2: (a) kernel-trace timeline with the per-queue dump; (b) bulk hash kernels as capped (persistent) grids.
set -u
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
V=tools/build/variants
rm -rf $OUT/_tl
rocprofv3 --kernel-trace --output-format csv -d $OUT/_tl -- python3 tools/prove_one.py 192 > $OUT/_tl.log 2>&1
find $OUT/_tl -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/timeline.py {} --dump $OUT/r05_d_trace_batch192.csv.gz > $OUT/r05_d_timeline_batch192.txt 2>&1
rm -rf $OUT/_tl
head -12 $OUT/r05_d_timeline_batch192.txt
P=$V/libp25_persist.so
python tools/ab_bench.py --rounds 2 --steps 3 base=base p0=$P p6144=$P@P25_X_HASH_GRID=6144 p4096=$P@P25_X_HASH_GRID=4096 \
  p3072=$P@P25_X_HASH_GRID=3072 p2048=$P@P25_X_HASH_GRID=2048 p1024=$P@P25_X_HASH_GRID=1024 > $OUT/r05_d_ab_persistent_hash.txt 2>&1
cut -c1-110 $OUT/r05_d_ab_persistent_hash.txt
