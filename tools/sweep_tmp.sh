cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt1; mkdir -p gpurun_out/kt1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/kt1/bench.json 2> gpurun_out/kt1/err.log
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/kt1/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
for t in 8 16 24; do for b in 3 4; do echo -n "thr=$t b=$b "; SMHIP_SCAN_THRESHOLD=$t SMHIP_LANE_BLOCKS_PER_CU=$b timeout 200 python bench.py --no-cpu --steps 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['kernels']['seed_lookup']['mean_ms'], d['config']['walker_phase_stats'])"; done; done
