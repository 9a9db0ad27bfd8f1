#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes (one counter set each, never with sys/hip/hsa tracing)
# over tools/micro/calib_traffic (known byte counts, walker access shapes) and over the bench's
# walker, raw output under gpurun_out/calib/.  tools/calib_summarize.py writes the committed
# profiles/r3_calibration.json from it.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/calib_traffic tools/micro/calib_traffic.hip || exit 1
$OUT/calib_traffic > $OUT/known_bytes.json || exit 1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
         "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum"; do
  i=$((i+1))
  echo "$c" > $OUT/pass_$i.counters
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/micro_$i -- $OUT/calib_traffic > /dev/null 2> $OUT/micro_${i}_err.log
  # (MICRO_ONLY=1: only the calibration kernels -- the bench's walker counters come from tools/collect_profiles.sh)
  [ "${MICRO_ONLY:-0}" = 1 ] || timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/bench_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --depth 1 > $OUT/bench_$i.json 2> $OUT/bench_${i}_err.log
done
du -sh $OUT
python3 tools/calib_summarize.py > $OUT/summary_print.txt 2>&1
# keep the merge-back small: raw per-dispatch CSVs of the bench passes are large
find $OUT -name '*counter_collection.csv' -size +8M -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
