#!/bin/bash
# round 5, call I: new download path (tests), final streaming numbers, GPU busy share of a streaming run (kernel trace)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_chain_stage.py -m gpu -x -q -k "carry or cli or chain or anchor" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
STREAM_ROUNDS_OUT=$O/rounds_c8_s2.txt timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 96 --workdir $W --rocprof-dir $O/kt_c8 > $O/stream_c8_s2.json 2> $O/stream_c8_s2.err; echo "c8 s2 rc=$?"; tail -c 200 $O/stream_c8_s2.err
STREAM_ROUNDS_OUT=$O/rounds_c30_s2.txt timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 2 --check 64 --workdir $W --rocprof-dir $O/kt_c30 > $O/stream_c30_s2.json 2> $O/stream_c30_s2.err; echo "c30 rc=$?"; tail -c 200 $O/stream_c30_s2.err
timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_50k_s2.json 2> $O/stream_c8_50k_s2.err
timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 0 --sdtw-band 64 --workdir $W > $O/stream_c8_s2_sdtw.json 2> $O/stream_c8_s2_sdtw.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5i/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], "used", j["mean_chunks_used"], "twice", j.get("batches_mapped_twice"), j.get("oracle_check"))
    print("    ", j["lane_seconds"])
PY
for T in c8 c30; do
  S=$(python3 -c "import json;print(json.loads(open('$O/stream_${T}_s2.json').read().strip().splitlines()[-1])['span_s']*1000)")
  python3 tools/timeline.py $O/kt_$T --last-ms $S --out $O/timeline_stream_$T.csv > $O/timeline_stream_$T.txt 2>&1; head -12 $O/timeline_stream_$T.txt
  find $O/kt_$T -name '*.csv' -size +2M -delete; find $O/kt_$T -name '*.db' -delete
done
