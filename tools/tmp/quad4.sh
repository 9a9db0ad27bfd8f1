run() { python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['census_results_identical'], d['config']['walker_phase_stats'])"; }
run f4_s8
SMHIP_BINARY_SET=1 run binary
for v in "3 9" "2 10" "3 8"; do
  set -- $v
  rm -f build/seed_lookup.o; make sigmap_amd/lib/libsigmap_hip.so EXTRA_HIPFLAGS="-DSMHIP_QUAD_STACK=$2 -DSMHIP_FIFO=$1" > /dev/null 2>&1
  run f$1_s$2
done
rm -f build/seed_lookup.o; make sigmap_amd/lib/libsigmap_hip.so > /dev/null 2>&1
