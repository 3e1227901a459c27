#!/bin/bash
# HBM traffic and VALU instructions per kernel at BASELINE config 5's size (outer circuit 2^19 rows, LDE 2^22: an 8x
# larger working set than fib-64) -- separate --pmc passes, each with --kernel-trace only.   tools/pmc_config5.sh r03_z
set -u
TAG=${1:-rXX}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# rocprofv3's preloaded tool library opens the GPU runtime before the program starts: a GPU_MAX_HW_QUEUES that libp25 sets at its
# own first call may come too late then (p25_runtime_info: hw_queues_setting_late) and the prover would run its 16 streams on
# the runtime's default 4 hardware queues -- exported here, so every pass is taken in the regime the product runs in.
export GPU_MAX_HW_QUEUES=24
for C in "SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/_pmc5
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc5 -- python3 tools/prove_one.py 2 --log-n 20 > $OUT/_pmc5.log 2>&1
  python3 tools/pmc_summary.py $OUT/_pmc5 $OUT/${TAG}_config5_pmc_${C}.json 2 > $OUT/${TAG}_config5_pmc_${C}.txt
  rm -rf $OUT/_pmc5
done
for f in $OUT/${TAG}_config5_pmc_*.txt; do tail -n 1 $f; done
