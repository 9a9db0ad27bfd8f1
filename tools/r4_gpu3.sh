#!/bin/bash
# round 4, GPU call 3: whole-line sort partition -- parity suite, then A/B against the exact two-pass route
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4_gpu3_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4_gpu3_tests.log
export SMHIP_DEV=1
for L in 1 0; do
  SMHIP_SORT_LINES=$L timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_yeast_lines$L.json 2> gpurun_out/r4_yeast_lines$L.err
  SMHIP_SORT_LINES=$L timeout 600 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 3 --warmup 1 --no-cpu --no-pcie --no-same-reads \
     > gpurun_out/r4_chr1_lines$L.json 2> gpurun_out/r4_chr1_lines$L.err
done
python tools/benchsum.py gpurun_out/r4_yeast_lines1.json gpurun_out/r4_yeast_lines0.json gpurun_out/r4_chr1_lines1.json gpurun_out/r4_chr1_lines0.json 2>&1 | tail -40
tail -3 gpurun_out/r4_chr1_lines1.err
