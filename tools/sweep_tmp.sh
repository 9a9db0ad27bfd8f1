for cl in 7500 75000 250000 750000 2000000; do
  echo -n "contig_len=$cl: "
  timeout 300 python bench.py --no-cpu --steps 3 --contig-len $cl 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], c['kernels']['seed_lookup']['mean_ms'], c['nodes_per_query'], c['points_per_query'], c['queries_per_chunk'], c['anchors_per_chunk'])"
done
