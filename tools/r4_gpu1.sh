#!/bin/bash
# round 4, GPU call 1: the whole GPU suite (new parity tests included), then baselines at 250 Mb / 8 000 chunks
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu1_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu1_tests.log
for fc in 512 1024 2048; do
  timeout 600 python bench.py --contigs 1 --contig-len 250000000 --reads 8000 --depth 1 --steps 3 --warmup 1 --no-cpu --no-pcie --fast-cap $fc \
    > gpurun_out/r4_chr1_8000_fc$fc.json 2> gpurun_out/r4_chr1_8000_fc$fc.err
  echo "chr1 fc=$fc rc=$?"
done
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu > gpurun_out/r4_yeast_base.json 2> gpurun_out/r4_yeast_base.err
echo "yeast rc=$?"
python tools/benchsum.py gpurun_out/r4_chr1_8000_fc512.json gpurun_out/r4_chr1_8000_fc1024.json gpurun_out/r4_chr1_8000_fc2048.json gpurun_out/r4_yeast_base.json 2>&1 | tail -40
