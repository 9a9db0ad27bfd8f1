timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py 40 2>&1 | tail -2
run() { python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['walker_nodes_per_query'], d['config']['census_results_identical'], d['config']['walker_phase_stats'])"; }
SMHIP_BINARY_SET=1 run binary
run quad7
SMHIP_QUAD_BLOCKS_PER_CU=6 run quad6
SMHIP_QUAD_BLOCKS_PER_CU=5 run quad5
SMHIP_QUAD_BLOCKS_PER_CU=8 run quad8
