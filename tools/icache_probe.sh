#!/bin/bash
# Instruction-cache behaviour of the proving kernels (SQC_ICACHE_* counters), per kernel: icache_probe.sh <tag>
set -u
TAG=${1:-icache}
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# rocprofv3's preloaded tool library opens the GPU runtime before the program starts: a GPU_MAX_HW_QUEUES that libp25 sets at its
# own first call may come too late then (p25_runtime_info: hw_queues_setting_late) and the prover would run its 16 streams on
# the runtime's default 4 hardware queues -- exported here, so every pass is taken in the regime the product runs in.
export GPU_MAX_HW_QUEUES=24
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL_[A-Z_]*\|SQC_TC_INST[A-Z_]*" | sort -u > $OUT/${TAG}_counters.txt
cat $OUT/${TAG}_counters.txt | tr '\n' ' '; echo
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  rm -rf $OUT/_pmc
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py 4 > $OUT/_pmc.log 2>&1
  python3 tools/pmc_by_kernel.py $OUT/_pmc k_ >> $OUT/${TAG}.txt 2>&1
  tail -3 $OUT/_pmc.log >> $OUT/${TAG}_log.txt
  rm -rf $OUT/_pmc
done
grep "k_hash_leaves\|k_quotient\|k_ntt_tile\|k_tree_level " $OUT/${TAG}.txt | cut -c1-260
