#!/bin/bash
# Builds libp25 variants that differ only in kernels_hash.hip's experiment switches (tools/exp/switches_kernels_hash.patch),
# into tools/build/variants/.
# usage: tools/hash_variants.sh name "-DP25_LEAF_MX=0 ..." [name2 "flags2" ...]
"$(dirname "$0")/variants.sh" kernels_hash.hip "$@"
