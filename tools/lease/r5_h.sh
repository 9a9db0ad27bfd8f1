#!/bin/bash
# round 5, call H: full GPU suite after the walker's adaptive hand-out + streaming numbers
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
timeout 1200 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log; tail -3 $O/suite.log
STREAM_ROUNDS_OUT=$O/rounds_c8_s2.txt timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 96 --workdir $W > $O/stream_c8_s2.json 2> $O/stream_c8_s2.err; echo "c8 s2 rc=$?"; tail -c 200 $O/stream_c8_s2.err
STREAM_ROUNDS_OUT=$O/rounds_c30_s3.txt timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 3 --check 64 --workdir $W > $O/stream_c30_s3.json 2> $O/stream_c30_s3.err; echo "c30 rc=$?"; tail -c 200 $O/stream_c30_s3.err
timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 2 --check 0 --workdir $W > $O/stream_c30_s2.json 2> $O/stream_c30_s2.err
timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_50k_s2.json 2> $O/stream_c8_50k_s2.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5h/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], "used", j["mean_chunks_used"], "spec", j.get("speculative_batches"), "twice", j.get("batches_mapped_twice"), j.get("oracle_check"))
PY
tail -8 $O/rounds_c8_s2.txt | cut -c1-200
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --repeats 3 > $O/bench.json 2> $O/bench.err; python3 tools/benchsum.py $O/bench.json | head -4
