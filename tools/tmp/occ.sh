for b in 4 3 2; do
SMHIP_LANE_BLOCKS_PER_CU=$b python bench.py --no-cpu --no-pcie --depth 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocks/CU', $b, d['ms_per_step'], d['serial_steps']['kernels']['seed_lookup'])"
done
