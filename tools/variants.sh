#!/bin/bash
# Builds libp25 variants that differ only in ONE translation unit's experiment switches, into tools/build/variants/.
# usage: tools/variants.sh kernels_quotient.hip name "-DP25_Q_WAVES=3" [name2 "flags2" ...]
# (tools/ab_bench.py runs the batch-256 proving loop on each of them, back to back on one box.)
set -e
SRC=$1; shift
STEM=${SRC%.*}
cd "$(dirname "$0")/../plonky2.5_amd/csrc"
make -s -j8
mkdir -p ../../tools/build/variants
OTHERS=$(ls build/*.o | grep -v "build/$STEM.o")
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include $FLAGS -x hip -c $SRC -o ../../tools/build/variants/${STEM}_$NAME.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/build/variants/libp25_$NAME.so $OTHERS ../../tools/build/variants/${STEM}_$NAME.o
  echo built $NAME
done
