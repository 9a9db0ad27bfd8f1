run() { python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['census_results_identical'], d['config']['walker_phase_stats'])"; }
run base
SMHIP_SCAN_THRESHOLD=2 run thr2
SMHIP_SCAN_THRESHOLD=8 run thr8
SMHIP_SCAN_THRESHOLD=16 run thr16
bash tools/phase_prof.sh run | head -1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
