#!/bin/bash
# Round 5, experiment 7: with the hash kernels as persistent grids the queues' pipes are no longer held by their dispatch
# (nine kernels execute at once instead of five): do MORE proofs in flight pay then?  streams 16 / 20 / 24, 24 or 32 queues.
set -u
OUT=gpurun_out
P=tools/build/variants/libp25_persist.so
python tools/ab_bench.py --rounds 1 --steps 3 base=base p16=$P@P25_X_HASH_GRID=6144 p20=$P@P25_X_HASH_GRID=6144,P25_AB_STREAMS=20 \
  p24=$P@P25_X_HASH_GRID=6144,P25_AB_STREAMS=24 b20=base@P25_AB_STREAMS=20 > $OUT/r05_o_ab_persistent_hash_streams.txt 2>&1
python tools/ab_bench.py --hwq 32 --rounds 1 --steps 3 base=base p20=$P@P25_X_HASH_GRID=6144,P25_AB_STREAMS=20 \
  p24=$P@P25_X_HASH_GRID=6144,P25_AB_STREAMS=24 p24g4096=$P@P25_X_HASH_GRID=4096,P25_AB_STREAMS=24 >> $OUT/r05_o_ab_persistent_hash_streams.txt 2>&1
cut -c1-100 $OUT/r05_o_ab_persistent_hash_streams.txt
