set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_${i}_err.log
  tail -2 $OUT/pmc_${i}_err.log
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcx/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f)):
        if 'SeedWalk' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); 
    disp = set()
    n = 0
    for r in csv.DictReader(open(f)):
        if 'SeedWalk' in r['Kernel_Name']: disp.add(r['Dispatch_Id'])
    for k,v in acc.items(): print(f.split('/')[2], k, "%.4g" % (v[0]/max(len(disp),1)))
PY
