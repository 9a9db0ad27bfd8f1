#!/bin/bash
# bench.py over index sizes (total bases = 16 x contig length); documentation table in DESIGN.md
for cl in 75000 287500 750000 2000000 6250000; do
  echo -n "contig_len=$cl: "
  timeout 900 python bench.py --no-cpu --steps 3 --contig-len $cl 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(d['value'], 'ms/step', d['ms_per_step'], 'seed_ms', c['kernels']['seed_lookup']['mean_ms'], 'frac', d['roofline']['frac'], 'Nn', c['nodes_per_query'], 'Np', c['points_per_query'], 'walker', c['walker_nodes_per_query'], c['walker_points_per_query'], 'A', c['anchors_per_chunk'], 'mapped', c['mapped_fraction'], 'flags', c['chunks_with_capacity_flags'], 'build_s', c['index_build_s'])"
done
