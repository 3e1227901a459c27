#!/bin/bash
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24   # the profiler opens the GPU runtime first: libp25's own request would come too late
# needs the profiling build: tools/exp/build_gatemask.sh  (the shipped libp25.so has no gate mask)
LIB=$PWD/tools/build/libp25_gatemask.so
for M in 0xFFFFFFFF 0x0 $*; do
  rm -rf gpurun_out/_pmc
  P25_Q_MASK=$M rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/_pmc -- python3 tools/prove_one.py 1 --lib $LIB > gpurun_out/_pmc.log 2>&1
  echo "MASK $M: $(python3 tools/pmc_summary.py gpurun_out/_pmc | grep k_quotient)"
done
rm -rf gpurun_out/_pmc
