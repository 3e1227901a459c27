#!/bin/bash
# Builds libp25 variants that differ only in ONE translation unit's experiment switches, into tools/build/variants/.
# The switches are not in the product sources: the translation unit is compiled from a scratch copy with the patches of
# tools/exp/switches_*.patch applied (tools/exp/apply.sh); every other object is the product's.
# usage: tools/variants.sh kernels_quotient.hip name "-DP25_Q_WAVES=3" [name2 "flags2" ...]
# (tools/ab_bench.py runs the batch-256 proving loop on each of them, back to back on one box.)
set -e
SRC=$1; shift
STEM=${SRC%.*}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$ROOT/plonky2.5_amd/csrc"
X=$("$ROOT/tools/exp/apply.sh" variants tools/exp/switches_kernels_hash.patch tools/exp/switches_kernels_ntt.patch \
    tools/exp/switches_kernels_quotient.patch tools/exp/switches_arith_sched.patch)
OUT=$ROOT/tools/build/variants
mkdir -p "$OUT"
OTHERS=$(ls "$ROOT"/plonky2.5_amd/csrc/build/*.o | grep -v "build/$STEM.o")
cd "$X"
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I../../tools $FLAGS -x hip -c $SRC -o "$OUT/${STEM}_$NAME.o"
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libp25_$NAME.so" $OTHERS "$OUT/${STEM}_$NAME.o"
  echo built $NAME
done
