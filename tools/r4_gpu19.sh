#!/bin/bash
# round 4, GPU call 19: set-walk workgroups per CU (22 KB of LDS each: seven fit), and what a kernel's serial time is worth in flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
export SMHIP_DEV=1
for sb in 8 7 6; do
  SMHIP_SET_BLOCKS=$sb timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 16 --warmup 4 > gpurun_out/r4_sb$sb.json 2> gpurun_out/r4_sb$sb.err
done
timeout 600 python bench.py --no-cpu --no-pcie --no-same-reads --steps 16 --warmup 4 --sdtw-band 0 > gpurun_out/r4_nosdtw.json 2> gpurun_out/r4_nosdtw.err
python tools/benchsum.py gpurun_out/r4_sb8.json gpurun_out/r4_sb7.json gpurun_out/r4_sb6.json gpurun_out/r4_nosdtw.json | grep -E "value|serial|flight"
