set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_${i}_err.log
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcx/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(f)):
        if 'SeedWalk' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    for k,v in acc.items(): print(f.split('/')[2], k, "%.4g" % (v/max(len(disp),1)))
PY
