#!/bin/bash
# round 4, GPU call 29: the size table on the final code
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
B="--no-cpu --no-pcie --no-same-reads"
timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 4 --warmup 2 $B > gpurun_out/r4f_chr1.json 2> gpurun_out/r4f_chr1.err
timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 4000 --depth 2 --steps 4 --warmup 2 $B > gpurun_out/r4f_chr1_4000x2.json 2> gpurun_out/r4f_chr1_4000x2.err
timeout 900 python bench.py --contigs 16 --contig-len 2000000 --depth 3 --steps 6 --warmup 3 $B > gpurun_out/r4f_32mb.json 2> gpurun_out/r4f_32mb.err
timeout 900 python bench.py --contigs 16 --contig-len 6250000 --depth 2 --steps 4 --warmup 2 $B > gpurun_out/r4f_100mb.json 2> gpurun_out/r4f_100mb.err
timeout 900 python bench.py --contigs 16 --contig-len 75000 --depth 4 --steps 8 --warmup 4 $B > gpurun_out/r4f_1mb.json 2> gpurun_out/r4f_1mb.err
timeout 900 python bench.py --contigs 1 --contig-len 4600000 --depth 4 --steps 8 --warmup 4 $B > gpurun_out/r4f_ecoli.json 2> gpurun_out/r4f_ecoli.err
timeout 900 python bench.py --steps 16 --warmup 4 $B > gpurun_out/r4f_yeast.json 2> gpurun_out/r4f_yeast.err
python tools/benchsum.py gpurun_out/r4f_1mb.json gpurun_out/r4f_ecoli.json gpurun_out/r4f_yeast.json gpurun_out/r4f_32mb.json gpurun_out/r4f_100mb.json gpurun_out/r4f_chr1.json gpurun_out/r4f_chr1_4000x2.json | grep -E "value|serial"
