#!/bin/bash
# round 4, GPU call 14: the chain classifier's dense-pair route: parity (chain stage, whole path, fuzz), then timing A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_chain_stage.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r4_gpu14_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu14_tests.log
export SMHIP_DEV=1
for d in -1 0; do
  export SMHIP_CLASSIFY_DENSE=$d
  bash tools/kstats.sh dense$d 2>&1 | grep -E "Classify|ChainDp|Pending"
  python tools/benchsum.py gpurun_out/kstats_dense$d/bench.json | grep -E "value|serial"
done
SMHIP_CLASSIFY_DENSE=-1 bash tools/kstats.sh dense_chr1 --contigs 1 --contig-len 250000000 --reads 4000 2>&1 | grep -E "Classify|ChainDp|Pending"
SMHIP_CLASSIFY_DENSE=0 bash tools/kstats.sh dense0_chr1 --contigs 1 --contig-len 250000000 --reads 4000 2>&1 | grep -E "Classify|ChainDp|Pending"
