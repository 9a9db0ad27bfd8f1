#!/bin/bash
# round 5, call K: counter profiles of the E. coli-sized and chr1-sized references (short sets) + default bench of each
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
bash tools/collect_profiles.sh r5_ecoli short --contigs 1 --contig-len 4600000
tail -5 gpurun_out/prof_r5_ecoli/summary_print.txt
timeout 600 python3 bench.py --contigs 1 --contig-len 4600000 --no-cpu --repeats 3 > $O/bench_ecoli.json 2> $O/bench_ecoli.err; python3 tools/benchsum.py $O/bench_ecoli.json | head -3
bash tools/collect_profiles.sh r5_chr1 short --contigs 1 --contig-len 250000000 --reads 8000
tail -5 gpurun_out/prof_r5_chr1/summary_print.txt
timeout 1500 python3 bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 6 --warmup 2 --no-cpu --no-pcie --no-same-reads --repeats 3 > $O/bench_chr1.json 2> $O/bench_chr1.err; python3 tools/benchsum.py $O/bench_chr1.json | head -4
