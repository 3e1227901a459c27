#!/bin/bash
# Round 5, experiment 3: the memory-waiting bulk kernels (NTT, quotient) as capped persistent grids -- a launch that fits the
# chip is dispatched at once (its queue's pipe is free) and holds a bounded share of every CU; the hash kernels keep theirs.
set -u
OUT=gpurun_out
mkdir -p $OUT
V=tools/build/variants
P=$V/libp25_persist2.so
# grids: quotient blocks of 128 lanes (2 waves): 1024 = 4 per CU = 2 waves/SIMD; NTT blocks of 256 lanes: 512 = 2 per CU
python tools/ab_bench.py --rounds 2 --steps 3 base=base pp0=$P \
  q2048=$P@P25_X_Q_GRID=2048 q1024=$P@P25_X_Q_GRID=1024 q512=$P@P25_X_Q_GRID=512 \
  n768=$P@P25_X_NTT_GRID=768 n512=$P@P25_X_NTT_GRID=512 n256=$P@P25_X_NTT_GRID=256 \
  q1024n512=$P@P25_X_Q_GRID=1024,P25_X_NTT_GRID=512 q2048n768=$P@P25_X_Q_GRID=2048,P25_X_NTT_GRID=768 > $OUT/r05_f_ab_persistent_ntt_quotient.txt 2>&1
cut -c1-110 $OUT/r05_f_ab_persistent_ntt_quotient.txt
