run() { python bench.py --no-cpu --no-pcie --steps 3 --depth 1 --contigs 1 --contig-len 250000000 --reads 2000 $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['serial_steps']; k=s['kernels']; print('$1', d['value'], d['ms_per_step'], 'seed', k['seed_lookup']['mean_ms'], 'sort', k['sort']['mean_ms'], 'chain', k['chain']['mean_ms'], d['config']['census_results_identical'], d['config']['overflow_queries'])"; }
run newovf ""
SMHIP_EXACT_NOSTOP=1 run newovf_nostop ""
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
