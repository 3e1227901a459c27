#!/bin/bash
# Builds tools/build/variants/libp25_<name>.so: the three bulk kernel files from a scratch copy of the product sources with the
# experiment switches patched back in (tools/exp/apply.sh + tools/exp/switches_*.patch) and -DP25_EXPERIMENT_KNOBS, i.e. with
# the environment-driven knobs compiled in (P25_X_Q_LDS_PAD, P25_X_NTT_LDS_PAD, P25_X_*_GRID).  The shipped library has neither
# the switches nor the knobs.  Extra arguments are appended to the compile flags.
# usage: tools/knobs_build.sh [name [extra flags]]     ->  tools/build/variants/libp25_<name>.so   (default name: knobs)
set -e
NAME=${1:-knobs}; shift || true
EXTRA="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$ROOT/plonky2.5_amd/csrc"
X=$("$ROOT/tools/exp/apply.sh" knobs_$NAME tools/exp/switches_kernels_hash.patch tools/exp/switches_kernels_ntt.patch \
    tools/exp/switches_kernels_quotient.patch tools/exp/switches_arith_sched.patch)
OUT=$ROOT/tools/build/variants
mkdir -p "$OUT/knobs_$NAME"
cd "$X"
for SRC in kernels_quotient.hip kernels_ntt.hip kernels_hash.hip; do
  STEM=${SRC%.*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DP25_EXPERIMENT_KNOBS $EXTRA -x hip -c $SRC -o "$OUT/knobs_$NAME/$STEM.o" &
done
wait
OTHERS=$(ls "$ROOT"/plonky2.5_amd/csrc/build/*.o | grep -v "build/kernels_quotient.o\|build/kernels_ntt.o\|build/kernels_hash.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libp25_$NAME.so" $OTHERS "$OUT"/knobs_$NAME/*.o
echo built "$OUT/libp25_$NAME.so"
