#!/bin/bash
# Round 5, experiment 6: k_quotient with the permutation argument inside the merged pass over the routed wires
# (P25_Q_MERGE_PERM: each routed wire column read once by the first two stages; 37 spilled VGPRs at 128).
set -u
OUT=gpurun_out
V=tools/build/variants
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python tools/exp/check_variant.py $V/libp25_qmerge.so > $OUT/r05_l_qmerge.txt 2>&1 || { tail -5 $OUT/r05_l_qmerge.txt; exit 1; }
python tools/ab_bench.py --rounds 2 --steps 3 base=base qmerge=$V/libp25_qmerge.so >> $OUT/r05_l_qmerge.txt 2>&1
for L in base $V/libp25_qmerge.so; do
  for C in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
    rm -rf $OUT/_pmc
    if [ $L = base ]; then rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py 4 > $OUT/_pmc.log 2>&1
    else rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py --lib $L 4 > $OUT/_pmc.log 2>&1; fi
    echo "== $L $C" >> $OUT/r05_l_qmerge.txt
    python3 tools/pmc_summary.py $OUT/_pmc $OUT/_x.json 4 | grep "k_quotient\|kernel " >> $OUT/r05_l_qmerge.txt
  done
done
rm -rf $OUT/_pmc
cut -c1-150 $OUT/r05_l_qmerge.txt
