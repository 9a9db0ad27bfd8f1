#!/bin/bash
# round 5, call J: CLI tests after the lane-level carry-over, streaming numbers, yeast profile set
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "carry or cli" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 96 --workdir $W > $O/stream_c8_s2.json 2> $O/stream_c8_s2.err; echo "c8 s2 rc=$?"; tail -c 200 $O/stream_c8_s2.err
timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 2 --check 64 --workdir $W > $O/stream_c30_s2.json 2> $O/stream_c30_s2.err; echo "c30 rc=$?"
timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_50k_s2.json 2> $O/stream_c8_50k_s2.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5j/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], j.get("oracle_check"))
    print("    ", j["lane_seconds"])
PY
bash tools/collect_profiles.sh r5_yeast full
cat gpurun_out/prof_r5_yeast/summary_print.txt | tail -30
