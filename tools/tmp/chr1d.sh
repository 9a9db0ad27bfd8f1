run() { python bench.py --no-cpu --no-pcie --steps 3 --depth 1 --contigs 1 --contig-len 250000000 --reads 2000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['serial_steps']; k=s['kernels']; print('$1', d['ms_per_step'], 'seed', k['seed_lookup']['mean_ms'], 'sort', k['sort']['mean_ms'], 'chain', k['chain']['mean_ms'], d['config']['walker_phase_stats'])"; }
SMHIP_SCAN_THRESHOLD=8 run thr8
SMHIP_SCAN_THRESHOLD=16 run thr16
