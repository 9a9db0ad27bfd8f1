#!/bin/bash
# round 4, GPU call 9: the counter profiles again (means over the launches of the batch's size), depth variants at 250 Mb
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
timeout 900 python bench.py --contigs 1 --contig-len 250000000 --reads 4000 --depth 2 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_chr1_4000x2.json 2> gpurun_out/r4_chr1_4000x2.err
timeout 900 python bench.py --contigs 16 --contig-len 2000000 --depth 3 --steps 6 --warmup 3 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_32mb.json 2> gpurun_out/r4_32mb.err
timeout 900 python bench.py --contigs 16 --contig-len 6250000 --depth 2 --steps 4 --warmup 2 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_100mb.json 2> gpurun_out/r4_100mb.err
timeout 900 python bench.py --contigs 16 --contig-len 75000 --depth 4 --steps 8 --warmup 4 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_1mb.json 2> gpurun_out/r4_1mb.err
timeout 900 python bench.py --contigs 1 --contig-len 4600000 --depth 4 --steps 8 --warmup 4 --no-cpu --no-pcie --no-same-reads > gpurun_out/r4_ecoli_d4.json 2> gpurun_out/r4_ecoli_d4.err
python tools/benchsum.py gpurun_out/prof_r4_yeast/bench_default.json gpurun_out/r4_chr1_4000x2.json gpurun_out/r4_32mb.json gpurun_out/r4_100mb.json gpurun_out/r4_1mb.json gpurun_out/r4_ecoli_d4.json 2>&1 | tail -40
