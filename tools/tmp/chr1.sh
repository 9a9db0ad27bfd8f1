run() { python bench.py --no-cpu --no-pcie --steps 4 --contigs 1 --contig-len 250000000 --reads 2000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['serial_steps']; k=s['kernels']; print('$1', d['value'], d['ms_per_step'], 'serial', s['ms_per_step'], 'seed', k['seed_lookup']['mean_ms'], 'sort', k['sort']['mean_ms'], 'chain', k['chain']['mean_ms'], d['config']['census_results_identical'], d['config']['contexts_agree'])"; }
run new
SMHIP_EXACT_NOSTOP=1 run nostop
