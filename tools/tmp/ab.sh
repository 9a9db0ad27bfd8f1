run() { python bench.py --no-cpu --no-pcie --depth 1 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup']['mean_ms'], d['config']['census_results_identical'])"; }
for i in 1 2; do
SMHIP_LIB=$PWD/sigmap_amd/lib/libsigmap_hip_base.so run base
run new
done
