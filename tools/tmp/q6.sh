timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('aos', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['census_results_identical'], d['config']['walker_phase_stats'])"
