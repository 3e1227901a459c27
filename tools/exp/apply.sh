#!/bin/bash
# Scratch copy of the product's kernel sources with experiment patches applied: the product tree itself never carries
# experiment switches (libp25.so builds from plonky2.5_amd/csrc with no -D flags).
# usage: tools/exp/apply.sh <scratch-name> <patch> [<patch> ...]   -> prints the scratch directory (tools/build/exp/<name>/csrc)
# The scratch tree keeps the layout (tools/build/exp/<name>/plonky2.5_amd/csrc, .../include) so that -I../../include works.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
DST=$ROOT/tools/build/exp/$NAME
rm -rf "$DST"
mkdir -p "$DST/plonky2.5_amd" "$DST/tools"
cp -r "$ROOT/include" "$DST/include"
mkdir -p "$DST/plonky2.5_amd/csrc"
cp "$ROOT"/plonky2.5_amd/csrc/*.{hip,cpp,h,inc} "$ROOT/plonky2.5_amd/csrc/Makefile" "$DST/plonky2.5_amd/csrc/"
cp "$ROOT/tools/poseidon_mfma.h" "$DST/tools/"
for P in "$@"; do
  case $P in /*) ;; *) P=$ROOT/$P ;; esac
  grep -v '^#' "$P" | patch -s -p1 -d "$DST"
done
echo "$DST/plonky2.5_amd/csrc"
