#!/bin/bash
# round 5, call M: co-residency again, now with 16 hardware queues: 6 walker workgroups per CU (walk-blocks 3) vs 8
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; rm -rf $O; mkdir -p $O
for WB in 0 3; do for D in 4 6; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-same-reads --repeats 3 --depth $D $([ $WB = 0 ] || echo --walk-blocks $WB) > $O/bench_wb${WB}_d$D.json 2> $O/bench_wb${WB}_d$D.err
  echo "walk-blocks=$WB depth=$D: $(python3 tools/benchsum.py $O/bench_wb${WB}_d$D.json | head -3 | tr '\n' ' ' | cut -c1-420)"
done; done
timeout 1500 python3 tools/fuzz_parity.py 300 52000 > $O/fuzz_300.log 2>&1; tail -2 $O/fuzz_300.log
