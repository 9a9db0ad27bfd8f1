#!/bin/bash
# round 4, GPU call 2: the whole GPU suite without -x (every failure listed)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4_gpu2_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r4_gpu2_tests.log
