#!/bin/bash
# round 4, GPU call 10: the whole suite on the final code, the default bench line (profile pairing), the CLI end to end
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4_gpu10_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r4_gpu10_tests.log
timeout 900 python bench.py > gpurun_out/r4_bench_default_final.json 2> gpurun_out/r4_bench_default_final.err
echo "bench rc=$?"
python tools/benchsum.py gpurun_out/r4_bench_default_final.json | tail -8
timeout 900 python tools/cli_e2e.py > gpurun_out/cli_e2e_r4.txt 2>&1
echo "cli rc=$?"; tail -12 gpurun_out/cli_e2e_r4.txt
