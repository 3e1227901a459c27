#!/bin/bash
# Round 5, experiment 8: pass 2 of the inverse NTT fused with pass 1 of the LDE (k_ntt_fused, P25_NTT_FUSE; n = 2^16: the
# fib-64 and the aggregation circuits).  Parity first, then the batch A/B, then the NTT's HBM traffic and wait share.
set -u
OUT=gpurun_out
V=tools/build/variants
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python tools/exp/check_variant.py $V/libp25_nttfuse.so > $OUT/r05_q_ntt_fused.txt 2>&1 || { tail -5 $OUT/r05_q_ntt_fused.txt; exit 1; }
python tools/ab_bench.py --rounds 2 --steps 3 base=base nttfuse=$V/libp25_nttfuse.so >> $OUT/r05_q_ntt_fused.txt 2>&1
for L in base $V/libp25_nttfuse.so; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    rm -rf $OUT/_pmc
    if [ $L = base ]; then rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py 4 > $OUT/_pmc.log 2>&1
    else rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py --lib $L 4 > $OUT/_pmc.log 2>&1; fi
    echo "== $L $C" >> $OUT/r05_q_ntt_fused.txt
    python3 tools/pmc_summary.py $OUT/_pmc $OUT/_x.json 4 | grep "k_ntt\|kernel \|TOTAL" >> $OUT/r05_q_ntt_fused.txt
  done
done
rm -rf $OUT/_pmc
cut -c1-160 $OUT/r05_q_ntt_fused.txt
