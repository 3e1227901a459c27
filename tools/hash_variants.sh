#!/bin/bash
# Builds libp25 variants that differ only in kernels_hash.hip's experiment switches, into tools/build/variants/.
# usage: tools/hash_variants.sh name "-DP25_LEAF_MX=0 ..." [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../plonky2.5_amd/csrc"
make -s -j8
mkdir -p ../../tools/build/variants
OTHERS=$(ls build/*.o | grep -v kernels_hash.o)
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I../../tools $FLAGS -c kernels_hash.hip -o ../../tools/build/variants/kernels_hash_$NAME.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/build/variants/libp25_$NAME.so $OTHERS ../../tools/build/variants/kernels_hash_$NAME.o
  echo built $NAME
done
