#!/bin/bash
# Round 5, experiment 1 (one gpurun call): (a) kernel-trace timeline of the batch pipeline; (b) occupancy caps A/B.
set -u
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
V=tools/build/variants
rm -rf $OUT/_tl
rocprofv3 --kernel-trace --output-format csv -d $OUT/_tl -- python3 tools/prove_one.py 192 > $OUT/_tl.log 2>&1
find $OUT/_tl -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/timeline.py {} > $OUT/r05_c_timeline_batch192.txt 2>&1
rm -rf $OUT/_tl
cat $OUT/r05_c_timeline_batch192.txt
python tools/ab_bench.py --rounds 2 --steps 3 base=base knobs=$V/libp25_knobs.so \
  q3=$V/libp25_knobs.so@P25_X_Q_LDS_PAD=13312 q2=$V/libp25_knobs.so@P25_X_Q_LDS_PAD=22528 q1=$V/libp25_knobs.so@P25_X_Q_LDS_PAD=51200 \
  n3=$V/libp25_knobs.so@P25_X_NTT_LDS_PAD=8192 n2=$V/libp25_knobs.so@P25_X_NTT_LDS_PAD=20480 \
  q2n2=$V/libp25_knobs.so@P25_X_Q_LDS_PAD=22528,P25_X_NTT_LDS_PAD=20480 \
  hash5=$V/libp25_hash5.so hash4=$V/libp25_hash4.so > $OUT/r05_c_ab_occupancy_caps.txt 2>&1
cut -c1-110 $OUT/r05_c_ab_occupancy_caps.txt
