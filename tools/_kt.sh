cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt_q; mkdir -p gpurun_out/kt_q
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_q -- python3 bench.py --steps 3 --warmup 1 --no-cpu --depth 1 > /dev/null 2> gpurun_out/kt_q/err.log
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/kt_q/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    print("%-60s %5s %10.1f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
