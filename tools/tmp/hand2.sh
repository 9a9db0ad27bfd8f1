run() { python bench.py --no-cpu --no-pcie --depth 1 --steps 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['serial_steps']['kernels']['seed_order']['mean_ms'], d['config']['census_results_identical'])"; }
for h in 1 2 4 8; do SMHIP_HANDOUT=$h run hand$h; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
