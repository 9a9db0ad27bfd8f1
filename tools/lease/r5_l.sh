#!/bin/bash
# round 5, call L: the new streaming test, pieces test, and the default bench exactly as the driver runs it (with the streaming block)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q -k "stream_bench or pieces" > $O/tests1.log 2>&1; tail -3 $O/tests1.log
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python3 tools/benchsum.py $O/bench.json | head -5
python3 -c "
import json
j=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print('streaming', j.get('streaming'))
print('cpu', j['cpu_baseline']['legs'])
print('roofline', {k:j['roofline'][k] for k in ('frac','traffic','useful_bytes_frac','step_useful_frac','launch_ms','frac_of_random_line_ceiling')})
"
