#!/bin/bash
# round 4, GPU call 17: sDTW sweep with fewer instructions per diagonal: refinement parity, timing; then the whole suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4_gpu17_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4_gpu17_tests.log
bash tools/kstats.sh sdtw2 2>&1 | grep -E "Sdtw|Classify"
python tools/benchsum.py gpurun_out/kstats_sdtw2/bench.json | grep -E "value|serial"
timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 16 --warmup 4 > gpurun_out/r4_gpu17_bench.json 2> gpurun_out/r4_gpu17_bench.err
timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_gpu17_chr1.json 2> gpurun_out/r4_gpu17_chr1.err
python tools/benchsum.py gpurun_out/r4_gpu17_bench.json gpurun_out/r4_gpu17_chr1.json | grep -E "value|serial|flight"
