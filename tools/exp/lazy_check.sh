#!/bin/bash
# Builds tools/build/variants/libp25_lazycheck.so -- the whole library from a scratch tree with tools/exp/lazy_contract_check.patch --
# and, on a GPU box, runs the variant parity script on it.   usage: tools/exp/lazy_check.sh [build|run]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if [ "${1:-build}" = build ]; then
  X=$("$ROOT/tools/exp/apply.sh" lazycheck tools/exp/lazy_contract_check.patch)
  make -s -j8 -C "$X"
  mkdir -p "$ROOT/tools/build/variants"
  cp "$X/../libp25.so" "$ROOT/tools/build/variants/libp25_lazycheck.so"
  echo built "$ROOT/tools/build/variants/libp25_lazycheck.so"
else
  cd "$ROOT" && python tools/exp/check_variant.py tools/build/variants/libp25_lazycheck.so
fi
