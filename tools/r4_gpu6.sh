#!/bin/bash
# round 4, GPU call 6: over-cap ranking + slices of 640 -- parity suite, then 250 Mb AUTO (set walk + ranking) vs EXACT
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0   # (a GPU memory fault otherwise fills the disk with a core dump)
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r4_gpu6_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4_gpu6_tests.log
for M in auto exact; do
  timeout 600 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 3 --warmup 1 --no-cpu --no-pcie --no-same-reads --search-mode $M \
     > gpurun_out/r4_chr1_v6_$M.json 2> gpurun_out/r4_chr1_v6_$M.err
done
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_yeast_v6.json 2> gpurun_out/r4_yeast_v6.err
python tools/benchsum.py gpurun_out/r4_chr1_v6_auto.json gpurun_out/r4_chr1_v6_exact.json gpurun_out/r4_yeast_v6.json 2>&1 | tail -30
