#!/bin/bash
# round 4, GPU call 16: batches in flight with fewer walker workgroups per CU (LDS left for the other contexts' kernels)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
for wb in 0 3 2; do
  timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --walk-blocks $wb --steps 16 --warmup 4 > gpurun_out/r4_wb$wb.json 2> gpurun_out/r4_wb$wb.err
done
for d in 3 5 6; do
  timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --depth $d --steps 16 --warmup 4 > gpurun_out/r4_depth$d.json 2> gpurun_out/r4_depth$d.err
done
python tools/benchsum.py gpurun_out/r4_wb0.json gpurun_out/r4_wb3.json gpurun_out/r4_wb2.json gpurun_out/r4_depth3.json gpurun_out/r4_depth5.json gpurun_out/r4_depth6.json | grep -E "value|serial|flight"
