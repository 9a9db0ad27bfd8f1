#!/bin/bash
# kernel-trace statistics of one serial bench run (GPU box): tools/kstats.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
OUT=gpurun_out/kstats_$tag
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --depth 1 "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))
if f:
    rows = list(csv.DictReader(open(f[-1])))
    for r in rows[:22]:
        n = r["Name"].replace("void ", "").replace("smhip::(anonymous namespace)::", "").split("(")[0]
        print("%-44s calls %5s  avg %10.1f us  total %8.2f ms  %5s%%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $OUT -name '*.csv' -size +4M -delete; find $OUT -name '*.db' -delete
