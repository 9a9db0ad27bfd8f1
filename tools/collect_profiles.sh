#!/bin/bash
# Runs on the GPU box (via gpurun): tools/collect_profiles.sh <tag> [full|short] [bench shape args...]
#   rocprofv3 kernel trace + stats of the bench command (batches in flight, and --depth 1 = standalone
#   kernel durations), then SEPARATE PMC passes (one counter set each, only ever with --kernel-trace: never
#   with sys / hip / hsa tracing), raw output under gpurun_out/prof_<tag>/.  tools/summarize_profiles.py <tag>
#   turns that into the committed profiles/<tag>_* files.
#   full  = all counter passes + the default bench run (the bench default shape);
#   short = the traffic and LDS passes only (the other BASELINE shapes).
set -u
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r4_yeast}; MODE=${2:-full}; shift; shift
SHAPE="$*"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
echo "$SHAPE" > $OUT/shape.txt
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --no-same-reads $SHAPE"
if [ "$MODE" = full ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt_bench.json 2> $OUT/kt_err.log
fi
# one batch in flight: standalone kernel durations
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_serial -- $BENCH --depth 1 > $OUT/kt_serial_bench.json 2> $OUT/kt_serial_err.log
PASSES_SHORT=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
PASSES_FULL=("TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
             "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM")
i=0
run_pass() {
  i=$((i+1))
  echo "$1" > $OUT/pmc_$i.counters
  timeout 1500 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --no-same-reads --depth 1 $SHAPE > /dev/null 2> $OUT/pmc_${i}_err.log
}
for c in "${PASSES_SHORT[@]}"; do run_pass "$c"; done
if [ "$MODE" = full ]; then
  for c in "${PASSES_FULL[@]}"; do run_pass "$c"; done
  timeout 900 python3 bench.py $SHAPE > $OUT/bench_default.json 2> $OUT/bench_default_err.log
fi
python3 tools/summarize_profiles.py $TAG --reduce-only > $OUT/summary_print.txt 2>&1
# keep the merge-back small: the reduced tables are in $OUT/reduced/
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*.db' -delete
du -sh $OUT
