run() { echo "$@"; timeout 300 python bench.py --no-cpu --steps 6 --warmup 1 --depth 1 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], {k:v['mean_ms'] for k,v in j['config']['kernels'].items()}, j['config']['overflow_queries'], j['config']['census_results_identical'])"; }
run --fast-cap 512
run --fast-cap 256
run --fast-cap 128
run --fast-cap 64
