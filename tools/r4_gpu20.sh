#!/bin/bash
# round 4, GPU call 20: anchors chained inside the classifier: parity (chain stage, whole path, fuzz, 250 Mb), timing A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_chain_stage.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r4_gpu20_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r4_gpu20_tests.log
export SMHIP_DEV=1
for d in -1 0; do
  export SMHIP_CLASSIFY_RESOLVE=$d
  bash tools/kstats.sh res$d 2>&1 | grep -E "Classify|ChainDp|Pending|ChainEnds"
  python tools/benchsum.py gpurun_out/kstats_res$d/bench.json | grep -E "value|serial"
done
SMHIP_CLASSIFY_RESOLVE=-1 bash tools/kstats.sh res_chr1 --contigs 1 --contig-len 250000000 --reads 4000 2>&1 | grep -E "Classify|ChainDp|Pending"
SMHIP_CLASSIFY_RESOLVE=0 bash tools/kstats.sh res0_chr1 --contigs 1 --contig-len 250000000 --reads 4000 2>&1 | grep -E "Classify|ChainDp|Pending"
