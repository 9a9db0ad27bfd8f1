cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pol
mkdir -p $OUT
timeout 120 $R/tools/micro/bin/random_policy > $OUT/rates.txt 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/pmc -- $R/tools/micro/bin/random_policy > /dev/null 2> $OUT/pmc_err.log
cat $OUT/rates.txt
find $OUT/pmc -name "*counter_collection.csv" | head -1 | xargs -I{} python3 -c "
import csv,collections,sys
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open('{}')):
    d[r['Kernel_Name']][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in d.items(): print(k[:60], dict(v))
"
