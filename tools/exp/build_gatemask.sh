#!/bin/bash
# Profiling-only library for tools/qmask.sh (per-gate instruction split of k_quotient): a scratch copy of the product sources
# with tools/exp/switches_kernels_quotient.patch applied, every translation unit built with -DP25_PROFILE_GATE_MASK (the
# switch adds a field to QuotientArgs), written to tools/build/libp25_gatemask.so -- outside the package, so it can never be
# loaded as the product.  It evaluates a SUBSET of the gates (env P25_Q_MASK) and so produces wrong proofs by design.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
X=$("$ROOT/tools/exp/apply.sh" gatemask tools/exp/switches_kernels_quotient.patch)
make -s -j8 -C "$X" CXXFLAGS_NORDC="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DP25_PROFILE_GATE_MASK"
mkdir -p "$ROOT/tools/build"
cp "$X/../libp25.so" "$ROOT/tools/build/libp25_gatemask.so"
echo built "$ROOT/tools/build/libp25_gatemask.so"
