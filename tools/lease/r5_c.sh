#!/bin/bash
# round 5, call C: piece uploads (new GPU test), CLI parity tests, streaming bench after the per-chunk uploads
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or cli or resident or streaming" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "cli" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
W=/tmp/sb; mkdir -p $W
for S in 2 3 4; do
  timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams $S --check $([ $S = 2 ] && echo 96 || echo 0) --workdir $W > $O/stream_c8_s$S.json 2> $O/stream_c8_s$S.err; echo "c8 s$S rc=$?"; tail -c 300 $O/stream_c8_s$S.err
done
GPU_MAX_HW_QUEUES=4 timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 3 --check 0 --workdir $W > $O/stream_c8_s3_q4.json 2> $O/stream_c8_s3_q4.err
timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 3 --check 64 --workdir $W > $O/stream_c30_s3.json 2> $O/stream_c30_s3.err; echo "c30 rc=$?"; tail -c 300 $O/stream_c30_s3.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], "used", j["mean_chunks_used"], j.get("oracle_check"))
    print("   ", j["lane_seconds"])
PY
