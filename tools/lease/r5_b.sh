#!/bin/bash
# round 5, call B: streaming-mode bench (first numbers) + HW-queue / depth sweep of the default bench
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
timeout 300 python3 tools/stream_bench.py --reads 600 --chunks-per-read 4 --check 24 --runs 1 --workdir $W > $O/stream_small.json 2> $O/stream_small.err; echo "small rc=$?"; tail -c 400 $O/stream_small.err
for S in 2 4; do
  timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams $S --check $([ $S = 2 ] && echo 96 || echo 0) --workdir $W > $O/stream_c8_s$S.json 2> $O/stream_c8_s$S.err; echo "c8 s$S rc=$?"
done
cat $O/stream_c8_s*.json
for Q in 0 8 16; do for D in 4 6; do
  if [ $Q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-same-reads --repeats 3 --depth $D > $O/bench_q${Q}_d$D.json 2> $O/bench_q${Q}_d$D.err
  echo "queues=$Q depth=$D: $(python3 tools/benchsum.py $O/bench_q${Q}_d$D.json | head -2 | tr '\n' ' ')"
done; done
