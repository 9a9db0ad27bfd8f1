#!/bin/bash
# tools/kstats_lib.sh <tag> <lib.so> <kernel-name-regex> [bench args]: kernel-trace averages of selected kernels with an
# alternative build of the device library (A/B experiments)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; lib=$2; pat=$3; shift 3
OUT=gpurun_out/kstats_$tag
rm -rf $OUT; mkdir -p $OUT
export SMHIP_LIB=$PWD/$lib
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --depth 1 "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - "$OUT" "$pat" "$tag" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))
if f:
    for r in csv.DictReader(open(f[-1])):
        n = r["Name"].replace("void ", "").replace("smhip::(anonymous namespace)::", "").split("(")[0]
        if re.search(sys.argv[2], n):
            print("%-10s %-40s calls %4s avg %9.1f us" % (sys.argv[3], n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
