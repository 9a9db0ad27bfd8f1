#!/bin/bash
# round 5, call D: streaming rounds trace, 50k-read streaming run, depth sweep at 16 queues, over-cap routes
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "repeat_injected or set_walk or pool" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
W=/tmp/sb; mkdir -p $W
STREAM_ROUNDS_OUT=$O/rounds_c8_s2.txt timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_s2.json 2> $O/stream_c8_s2.err; echo "c8 rc=$?"
STREAM_ROUNDS_OUT=$O/rounds_c8_50k_s2.txt timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_50k_s2.json 2> $O/stream_c8_50k_s2.err; echo "c8 50k rc=$?"
timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 3 --check 0 --workdir $W > $O/stream_c8_50k_s3.json 2> $O/stream_c8_50k_s3.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], "used", j["mean_chunks_used"])
    print("   ", j["lane_seconds"])
PY
cat $O/rounds_c8_s2.txt | cut -c1-230
for D in 4 5 6 8; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-same-reads --repeats 3 --depth $D > $O/bench_d$D.json 2> $O/bench_d$D.err
  echo "depth=$D: $(python3 tools/benchsum.py $O/bench_d$D.json | head -2 | tr '\n' ' ')"
done
