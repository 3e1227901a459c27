#!/bin/bash
# Builds tools/build/variants/libp25_knobs.so: the product's sources with -DP25_EXPERIMENT_KNOBS, i.e. with the
# environment-driven experiment knobs of kernels_quotient.hip / kernels_ntt.hip compiled in (P25_X_Q_LDS_PAD,
# P25_X_NTT_LDS_PAD).  The shipped library never reads these.  Extra arguments are appended to the compile flags.
# usage: tools/knobs_build.sh [name [extra flags]]     ->  tools/build/variants/libp25_<name>.so   (default name: knobs)
set -e
NAME=${1:-knobs}; shift || true
EXTRA="$*"
cd "$(dirname "$0")/../plonky2.5_amd/csrc"
make -s -j8
OUT=../../tools/build/variants
mkdir -p $OUT/knobs_$NAME
OBJS=""
for SRC in kernels_quotient.hip kernels_ntt.hip kernels_hash.hip; do
  STEM=${SRC%.*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DP25_EXPERIMENT_KNOBS $EXTRA -x hip -c $SRC -o $OUT/knobs_$NAME/$STEM.o &
done
wait
OTHERS=$(ls build/*.o | grep -v "build/kernels_quotient.o\|build/kernels_ntt.o\|build/kernels_hash.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libp25_$NAME.so $OTHERS $OUT/knobs_$NAME/*.o
echo built $OUT/libp25_$NAME.so
