#!/bin/bash
# round 4, GPU call 25: final checks -- build() + smoke(), the whole suite, the default bench line, the CLI end to end
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ulimit -c 0
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4_gpu25_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4_gpu25_tests.log
timeout 900 python bench.py > gpurun_out/r4_bench_default_final3.json 2> gpurun_out/r4_bench_default_final3.err
echo "bench rc=$?"
python tools/benchsum.py gpurun_out/r4_bench_default_final3.json | tail -6
timeout 900 python tools/cli_e2e.py > gpurun_out/cli_e2e_r4b.txt 2>&1
echo "cli rc=$?"; grep -E "streams|index" gpurun_out/cli_e2e_r4b.txt | cut -c1-200
