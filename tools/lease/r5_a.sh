#!/bin/bash
# round 5, call A: GPU suite on the changed code, default bench, kernel-trace timeline with four batches in flight
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log; tail -3 $O/suite.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 12 --warmup 4 --no-cpu --no-pcie --no-same-reads --repeats 2 > $O/kt_bench.json 2> $O/kt.err
python3 tools/timeline.py $O/kt --last-ms 230 --out $O/timeline_depth4.csv > $O/timeline_depth4.txt 2>&1
head -60 $O/timeline_depth4.txt
find $O/kt -name '*.csv' -size +2M -delete; find $O/kt -name '*.db' -delete
python3 tools/benchsum.py $O/bench.json 2>/dev/null | head -40
