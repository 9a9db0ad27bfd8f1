#!/bin/bash
# round 4, GPU call 24: events kernels with more values in flight per lane (prefix sums: kPre, features: kBlk): A/B + parity
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
bash tools/kstats_lib.sh ev_base sigmap_amd/lib/libsigmap_hip.so "Prefix|Feature"
for v in 32_8 16_16 32_16 48_16; do
  bash tools/kstats_lib.sh ev_$v sigmap_amd/lib/libsigmap_hip_ev_$v.so "Prefix|Feature"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "events or golden or pipeline or features" 2>&1 | tail -2
