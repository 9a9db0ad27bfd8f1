#!/bin/bash
# round 5, call G: streaming after the capacity fix (small batches use what the buffers already hold) + kernel trace of the tail rounds
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spec or resident or cli or pool" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
STREAM_ROUNDS_OUT=$O/rounds_c8_s2.txt timeout 900 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 96 --workdir $W > $O/stream_c8_s2.json 2> $O/stream_c8_s2.err; echo "c8 s2 rc=$?"; tail -c 200 $O/stream_c8_s2.err
STREAM_ROUNDS_OUT=$O/rounds_c30_s3.txt timeout 1200 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 30 --streams 3 --check 64 --workdir $W > $O/stream_c30_s3.json 2> $O/stream_c30_s3.err; echo "c30 rc=$?"; tail -c 200 $O/stream_c30_s3.err
timeout 1200 python3 tools/stream_bench.py --reads 50000 --chunks-per-read 8 --streams 2 --check 0 --workdir $W > $O/stream_c8_50k_s2.json 2> $O/stream_c8_50k_s2.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5g/stream_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], "reads/s", j["reads_per_s"], "chunks/s", j["chunks_per_s"], "span", j["all_span_s"], "rounds", j["rounds"], "mean_batch", j["mean_batch"],
          "host_share", j["host_share_of_lane_time"], "host_ms/round", j["host_ms_per_round"], "dev_ms/round", j["device_ms_per_round"], "used", j["mean_chunks_used"], "spec", j.get("speculative_batches"), "twice", j.get("batches_mapped_twice"), j.get("oracle_check"))
PY
cut -c1-200 $O/rounds_c8_s2.txt
SIGMAP_STREAM_STATS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_cli -- sigmap_amd/bin/sigmap -m -r $W/ref.fa -p $W/model.txt -x $W/idx -s $W/reads_r10000_c8 -o $W/out3.paf --streams 1 > $O/kt_cli.out 2> $O/kt_cli.err
grep STREAM_ROUND $O/kt_cli.err | tail -4 | cut -c1-200
python3 tools/timeline.py $O/kt_cli --last-ms 8 --out $O/timeline_tail.csv > $O/timeline_tail.txt 2>&1; head -50 $O/timeline_tail.txt
find $O/kt_cli -name '*.csv' -size +2M -delete; find $O/kt_cli -name '*.db' -delete
