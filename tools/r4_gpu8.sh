#!/bin/bash
# round 4, GPU call 8: counter profiles of the three BASELINE reference sizes on the final code, the default bench line,
# a 400-round fuzz log
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
bash tools/collect_profiles.sh r4_yeast full > gpurun_out/collect_r4_y.log 2>&1
echo "yeast done rc=$?"
bash tools/collect_profiles.sh r4_chr1 short --contigs 1 --contig-len 250000000 --reads 8000 > gpurun_out/collect_r4_c.log 2>&1
echo "chr1 done rc=$?"
bash tools/collect_profiles.sh r4_ecoli short --contigs 1 --contig-len 4600000 > gpurun_out/collect_r4_e.log 2>&1
echo "ecoli done rc=$?"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/fuzz_parity.py 400 1001 > gpurun_out/r4_fuzz_400_rounds.log 2>&1
echo "fuzz rc=$?"; tail -2 gpurun_out/r4_fuzz_400_rounds.log
python tools/benchsum.py gpurun_out/prof_r4_yeast/bench_default.json gpurun_out/prof_r4_yeast/kt_serial_bench.json gpurun_out/prof_r4_chr1/kt_serial_bench.json gpurun_out/prof_r4_ecoli/kt_serial_bench.json 2>&1 | tail -30
