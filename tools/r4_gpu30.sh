#!/bin/bash
# round 4, GPU call 30: the frozen code -- whole suite, counter profiles, default bench line, CLI end to end
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4_gpu30_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4_gpu30_tests.log
bash tools/collect_profiles.sh r4_yeast full > gpurun_out/collect_r4_y.log 2>&1
echo "yeast done rc=$?"
bash tools/collect_profiles.sh r4_chr1 short --contigs 1 --contig-len 250000000 --reads 8000 > gpurun_out/collect_r4_c.log 2>&1
echo "chr1 done rc=$?"
bash tools/collect_profiles.sh r4_ecoli short --contigs 1 --contig-len 4600000 > gpurun_out/collect_r4_e.log 2>&1
echo "ecoli done rc=$?"
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > gpurun_out/r4_bench_default_final4.json 2> gpurun_out/r4_bench_default_final4.err
echo "bench rc=$?"
python tools/benchsum.py gpurun_out/r4_bench_default_final4.json | tail -6
timeout 900 python tools/cli_e2e.py > gpurun_out/cli_e2e_r4c.txt 2>&1
echo "cli rc=$?"; grep -E "streams" gpurun_out/cli_e2e_r4c.txt | cut -c1-160
