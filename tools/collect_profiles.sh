#!/bin/bash
# Collects the evidence bundle of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02_z
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/ (tools/make_traffic_json.py <tag> rewrites
# profiles/pmc_hash_leaves.json from the FETCH/WRITE passes).
# A second argument selects sections (a gpurun call is limited to 20 minutes): tests | pmc | bench | agg | config5 (default: all).
# Order for a final bundle: pmc, agg and config5 first, copy the PMC json files into profiles/ (bench.py's VALU views read
# them and check the kernel sources' hash), then bench.
set -u
TAG=${1:-rXX}
WHAT=${2:-all}
case "$TAG" in -*|"") echo "usage: tools/collect_profiles.sh <tag> [tests|pmc|bench|agg|config5|micro ...]   (a tag names the output files: it cannot start with '-')" >&2; exit 2;; esac
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# rocprofv3's preloaded tool library opens the GPU runtime before the program starts: a GPU_MAX_HW_QUEUES that libp25 sets at its
# own first call may come too late then (p25_runtime_info: hw_queues_setting_late) and the prover would run its 16 streams on
# the runtime's default 4 hardware queues -- exported here, so every pass is taken in the regime the product runs in.
export GPU_MAX_HW_QUEUES=24
want() { [ "$WHAT" = all ] || [[ " $WHAT " == *" $1 "* ]]; }
if want tests; then
# (written as it runs: output held back by a pipe into tail makes a long run look hung to gpurun's silence detector)
python -m pytest tests -m gpu -q --durations=8 > $OUT/${TAG}_pytest_gpu.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/${TAG}_pytest_gpu.txt 2>&1
tail -3 $OUT/${TAG}_pytest_gpu.txt
fi
if want pmc; then
# PMC passes (each alone with --kernel-trace): a batch of 4 = the kernels of the throughput path
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf $OUT/_pmc
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_one.py 4 > $OUT/_pmc.log 2>&1
  python3 tools/pmc_summary.py $OUT/_pmc $OUT/${TAG}_pmc_${N}.json 4 > $OUT/${TAG}_pmc_${N}.txt
  rm -rf $OUT/_pmc
done
fi
if want bench; then
( time python bench.py ) > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
# rocprofv3 per-kernel summary of the same command (smaller batch) + the dominant kernel's launches split into
# "GPU to itself" (bench.py's single-proof passes = roofline.avg_launch_ms) and "in flight"
rm -rf $OUT/_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof -- python3 bench.py --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --extra-configs none --aggregate 0 > $OUT/${TAG}_bench_b64_rocprof.json 2> $OUT/_prof.err
find $OUT/_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_rocprofv3_kernel_stats_bench_b64.csv
find $OUT/_prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/rocprof_kernel_split.py {} > $OUT/${TAG}_rocprof_hash_leaves_split.txt
rm -rf $OUT/_prof
cut -c1-160 $OUT/${TAG}_bench_default.json
cat $OUT/${TAG}_rocprof_hash_leaves_split.txt
fi
if want agg; then
# the arity-8 aggregator: rocprofv3 per-kernel summary of the serial fold (64 leaves -> 8 -> 1), its throughput-form
# instruction counts (a batch of 8 level-1 proofs), and the pipelined device-resident tree
rm -rf $OUT/_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_prof -- python3 tools/aggregate.py 64 8 > $OUT/${TAG}_aggregate_64.json 2> $OUT/_agg.err
find $OUT/_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_rocprofv3_kernel_stats_aggregate_64_arity8.csv
rm -rf $OUT/_prof $OUT/_pmc
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/_pmc -- python3 tools/prove_agg.py 8 > $OUT/_pmc.log 2>&1
python3 tools/pmc_summary.py $OUT/_pmc $OUT/${TAG}_agg8_pmc_SQ_INSTS_VALU.json 8 > $OUT/${TAG}_agg8_pmc_SQ_INSTS_VALU.txt
rm -rf $OUT/_pmc
python tools/ab_bench.py --rounds 1 --steps 2 --pipe 4 base=base > $OUT/${TAG}_pipelined_tree.txt 2>&1
cat $OUT/${TAG}_pipelined_tree.txt | cut -c1-60,330-600
fi
if want config5; then
# BASELINE config 5: PMC passes at the same sources (bench.py's configs.config5.valu reads the SQ_INSTS_VALU file)
bash tools/pmc_config5.sh ${TAG} > $OUT/${TAG}_config5_run.log 2>&1
tail -3 $OUT/${TAG}_config5_run.log
fi
if want micro; then
tools/build/latbench > $OUT/${TAG}_latbench.txt 2>&1
tools/build/coopbench > $OUT/${TAG}_coopbench.txt 2>&1
python tools/kernel_bench.py > $OUT/${TAG}_kernel_bench.txt 2>&1
cat $OUT/${TAG}_kernel_bench.txt
fi
