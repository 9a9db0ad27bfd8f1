#!/bin/bash
# round 4, GPU call 12: the events kernels over LDS tiles (prefix sums, peak picker, features): parity, then timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r4_gpu12_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu12_tests.log
timeout 900 python bench.py --no-cpu --no-pcie > gpurun_out/r4_gpu12_bench.json 2> gpurun_out/r4_gpu12_bench.err
python tools/benchsum.py gpurun_out/r4_gpu12_bench.json | grep -E "value|serial|flight"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gpu12 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pcie --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_gpu12 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
