#!/bin/bash
# round 4, GPU call 22: the whole suite with anchors chained in the classifier; a long fuzz run; default + 250 Mb bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4_gpu22_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4_gpu22_tests.log
timeout 1500 python tools/fuzz_parity.py 300 7000 > gpurun_out/r4_fuzz_300_rounds_b.log 2>&1
echo "fuzz rc=$?"; tail -3 gpurun_out/r4_fuzz_300_rounds_b.log
timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 16 --warmup 4 > gpurun_out/r4_gpu22_bench.json 2> gpurun_out/r4_gpu22_bench.err
timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_gpu22_chr1.json 2> gpurun_out/r4_gpu22_chr1.err
timeout 900 python bench.py --contigs 16 --contig-len 6250000 --depth 2 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_gpu22_100mb.json 2> gpurun_out/r4_gpu22_100mb.err
python tools/benchsum.py gpurun_out/r4_gpu22_bench.json gpurun_out/r4_gpu22_chr1.json gpurun_out/r4_gpu22_100mb.json | grep -E "value|serial|flight"
