#!/bin/bash
# bench.py over index sizes (total bases = 16 x contig length, plus the 1-contig cases); documentation
# table in DESIGN.md.  Kernel times are those of the serial pass (one batch in flight).
row() {
  echo -n "$*: "
  timeout 1500 python bench.py --no-cpu --steps 4 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; s=d.get('serial_steps') or {'ms_per_step': d['ms_per_step'], 'kernels': c['kernels'], 'value': d['value']}
k=s['kernels']
print('value', d['value'], 'ms/step', d['ms_per_step'], '| serial', s['value'], s['ms_per_step'], 'seed', k['seed_lookup']['mean_ms'], 'sort', k['sort']['mean_ms'], 'chain', k['chain']['mean_ms'], 'sdtw', k['sdtw']['mean_ms'], 'frac', d['roofline']['frac'], '| Nn', c['nodes_per_query'], 'Np', c['points_per_query'], 'walker', c['walker_nodes_per_query'], c['walker_points_per_query'], 'A', c['anchors_per_chunk'], 'mapped', c['mapped_fraction'], 'flags', c['chunks_with_capacity_flags'], 'same', c['census_results_identical'], c['contexts_agree'], 'build_s', c['index_build_s'])"
}
for cl in 75000 750000 2000000 6250000; do row --contig-len $cl; done
row --contigs 1 --contig-len 4600000
if [ "${1:-}" = "chr1" ]; then row --contigs 1 --contig-len 250000000 --reads 2000; fi
