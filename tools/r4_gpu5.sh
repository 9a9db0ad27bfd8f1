#!/bin/bash
# round 4, GPU call 5: what the set walk alone costs at 250 Mb / 8 000 chunks (kernel trace), next to the exact walk
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for M in set exact; do
  rm -rf gpurun_out/kt_chr1_$M
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_chr1_$M -- python3 bench.py --contigs 1 --contig-len 250000000 \
     --reads 8000 --depth 1 --steps 2 --warmup 1 --no-cpu --no-pcie --no-same-reads --search-mode $M > gpurun_out/r4_chr1_mode_$M.json 2> gpurun_out/r4_chr1_mode_$M.err
  echo "mode $M rc=$?"
  f=$(ls gpurun_out/kt_chr1_$M/*/*kernel_stats.csv | tail -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("  %-70s calls %5s avg_us %10.1f total_ms %9.2f"%(r['Name'].replace('void smhip::(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
  find gpurun_out/kt_chr1_$M -name '*kernel_trace.csv' -delete; find gpurun_out/kt_chr1_$M -name '*.db' -delete
done
python tools/benchsum.py gpurun_out/r4_chr1_mode_set.json gpurun_out/r4_chr1_mode_exact.json | tail -12
