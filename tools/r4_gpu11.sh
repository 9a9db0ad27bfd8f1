#!/bin/bash
# round 4, GPU call 11: the whole-line partition with tiles of 4096 (one workgroup per CU) against tiles of 2048, A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
export SMHIP_DEV=1
for t in 2 4; do
  export SMHIP_SORT_LINES=$t
  timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_tile${t}_chr1.json 2> gpurun_out/r4_tile${t}_chr1.err
  timeout 900 python bench.py --contigs 16 --contig-len 6250000 --depth 2 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_tile${t}_100mb.json 2> gpurun_out/r4_tile${t}_100mb.err
  timeout 900 python bench.py --contigs 16 --contig-len 2000000 --depth 3 --steps 6 --warmup 3 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_tile${t}_32mb.json 2> gpurun_out/r4_tile${t}_32mb.err
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_tile${t}_yeast.json 2> gpurun_out/r4_tile${t}_yeast.err
done
python tools/benchsum.py gpurun_out/r4_tile2_chr1.json gpurun_out/r4_tile4_chr1.json gpurun_out/r4_tile2_100mb.json gpurun_out/r4_tile4_100mb.json gpurun_out/r4_tile2_32mb.json gpurun_out/r4_tile4_32mb.json gpurun_out/r4_tile2_yeast.json gpurun_out/r4_tile4_yeast.json 2>&1 | grep -E "value|serial" 
SMHIP_SORT_LINES=4 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r4_tile4_tests.log 2>&1
echo "tests(tile 4096) rc=$?"; tail -4 gpurun_out/r4_tile4_tests.log
