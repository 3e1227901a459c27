#!/bin/bash
# Round 5, experiment 4: does the in-pipeline inflation of the small multi-wave kernels (k_zpp_chunks 99 -> 689 us, k_pow_search
# 15 -> 209 us ...) come from the bulk hash kernels' single-wave workgroups winning every freed slot?  Timeline of the same
# batch with the hash kernels as persistent grids of 4096 / 3072 workgroups (their dispatch ends at once, 2-3 wave slots per
# SIMD stay free).
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=tools/build/variants/libp25_persist.so
for G in 4096 3072; do
  rm -rf $OUT/_tl
  P25_X_HASH_GRID=$G rocprofv3 --kernel-trace --output-format csv -d $OUT/_tl -- python3 tools/prove_one.py --lib $P 192 > $OUT/_tl.log 2>&1
  find $OUT/_tl -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/timeline.py {} > $OUT/r05_j_timeline_persist_hash_$G.txt 2>&1
  rm -rf $OUT/_tl
  head -4 $OUT/r05_j_timeline_persist_hash_$G.txt; tail -16 $OUT/r05_j_timeline_persist_hash_$G.txt
done
