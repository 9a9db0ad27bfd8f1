#!/bin/bash
# round 4, GPU call 27: leaf-order resolution again, interleaved repeats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
export SMHIP_DEV=1
for rep in 1 2; do
for bb in 0 15 14 13; do
  SMHIP_BUCKET_BITS=$bb timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 16 --warmup 4 > gpurun_out/r4_bb${bb}_$rep.json 2> gpurun_out/r4_bb${bb}_$rep.err
done
done
python tools/benchsum.py gpurun_out/r4_bb0_1.json gpurun_out/r4_bb15_1.json gpurun_out/r4_bb14_1.json gpurun_out/r4_bb13_1.json gpurun_out/r4_bb0_2.json gpurun_out/r4_bb15_2.json gpurun_out/r4_bb14_2.json gpurun_out/r4_bb13_2.json | grep -E "value"
