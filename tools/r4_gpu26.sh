#!/bin/bash
# round 4, GPU call 26: leaf-order resolution of the visiting order (query-key descent depth) against the walker's time
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
export SMHIP_DEV=1
for bb in 0 14 12 10; do
  SMHIP_BUCKET_BITS=$bb timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 12 --warmup 3 > gpurun_out/r4_bb$bb.json 2> gpurun_out/r4_bb$bb.err
done
python tools/benchsum.py gpurun_out/r4_bb0.json gpurun_out/r4_bb14.json gpurun_out/r4_bb12.json gpurun_out/r4_bb10.json | grep -E "value|serial"
