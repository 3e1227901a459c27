#!/bin/bash
# Round 5, experiment 5: BASELINE config 5 (2^19-row circuit, LDE 2^22) with the NTT / quotient as capped persistent grids.
set -u
OUT=gpurun_out
V=tools/build/variants
P=$V/libp25_persist2.so
python tools/ab_bench.py --config5 --rounds 1 --steps 3 base=base pp0=$P n1024=$P@P25_X_NTT_GRID=1024 n512=$P@P25_X_NTT_GRID=512 \
  q2048=$P@P25_X_Q_GRID=2048 n1024q2048=$P@P25_X_NTT_GRID=1024,P25_X_Q_GRID=2048 > $OUT/r05_k_ab_config5_persistent.txt 2>&1
cut -c1-140 $OUT/r05_k_ab_config5_persistent.txt
