#!/bin/bash
# round 5, call N: pack records A/B -- parity (set-walk tests with packs on and off; the parity suite with packs on), then time
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "set_walk" > $O/tests_setwalk.log 2>&1; tail -3 $O/tests_setwalk.log
SMHIP_PACKS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/tests_packs_on.log 2>&1; tail -3 $O/tests_packs_on.log
for P in 0 1; do
  SMHIP_DEV=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-same-reads --repeats 3 --packs $P > $O/bench_p$P.json 2> $O/bench_p$P.err
  echo "packs=$P: $(python3 tools/benchsum.py $O/bench_p$P.json | head -5 | tr '\n' ' ' | cut -c1-900)"
done
for P in 0 1; do
  SMHIP_DEV=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pcie --no-same-reads --repeats 3 --packs $P --contigs 1 --contig-len 4600000 > $O/bench_ecoli_p$P.json 2> $O/bench_ecoli_p$P.err
  echo "ecoli packs=$P: $(python3 tools/benchsum.py $O/bench_ecoli_p$P.json | head -3 | tr '\n' ' ' | cut -c1-600)"
done
