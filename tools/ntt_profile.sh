#!/bin/bash
# NTT-only A/B on the GPU box: ntt_profile.sh <tag> name=lib ...   (lib "base" = the product)
set -u
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# rocprofv3's preloaded tool library opens the GPU runtime before the program starts: a GPU_MAX_HW_QUEUES that libp25 sets at its
# own first call may come too late then (p25_runtime_info: hw_queues_setting_late) and the prover would run its 16 streams on
# the runtime's default 4 hardware queues -- exported here, so every pass is taken in the regime the product runs in.
export GPU_MAX_HW_QUEUES=24
for V in "$@"; do
  NAME=${V%%=*}; LIB=${V#*=}
  ARGS=""; [ "$LIB" != base ] && ARGS="--lib $LIB"
  echo "== $NAME" >> $OUT/${TAG}.txt
  python3 tools/ntt_only.py $ARGS --reps 16 >> $OUT/${TAG}.txt 2>&1
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    rm -rf $OUT/_pmc
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/ntt_only.py $ARGS --reps 1 > $OUT/_pmc.log 2>&1
    python3 tools/pmc_by_kernel.py $OUT/_pmc k_ntt >> $OUT/${TAG}.txt
    rm -rf $OUT/_pmc
  done
done
cat $OUT/${TAG}.txt
