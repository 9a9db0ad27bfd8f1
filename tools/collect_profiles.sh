#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the default bench command,
# then separate PMC passes (never combined with sys/hip/hsa tracing), raw output under
# gpurun_out/prof/.  tools/summarize_profiles.py turns that into the committed profiles/ files.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/kt_bench.json 2> $OUT/kt_err.log
# the same with one batch in flight: standalone kernel durations
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_serial -- $BENCH --depth 1 > $OUT/kt_serial_bench.json 2> $OUT/kt_serial_err.log
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
         "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-pcie --depth 1 > /dev/null 2> $OUT/pmc_${i}_err.log
done
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default_err.log
du -sh $OUT
