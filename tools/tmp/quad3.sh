run() { python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['walker_nodes_per_query'], d['config']['census_results_identical'])"; }
run stack8_b8
SMHIP_BINARY_SET=1 run binary
for st in 6 4; do
  rm -f build/seed_lookup.o; make sigmap_amd/lib/libsigmap_hip.so EXTRA_HIPFLAGS=-DSMHIP_QUAD_STACK=$st > /dev/null 2>&1
  run stack${st}_b8
done
rm -f build/seed_lookup.o; make sigmap_amd/lib/libsigmap_hip.so > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
