#!/bin/bash
# round 5, call E: what the first chunk round of a CLI context waits for (allocation / sync-route trace)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; rm -rf $O; mkdir -p $O
W=/tmp/sb; mkdir -p $W
[ -f $W/idx.pt ] || timeout 600 python3 tools/stream_bench.py --reads 10000 --chunks-per-read 8 --streams 2 --check 0 --runs 1 --workdir $W > $O/stream.json 2> $O/stream.err
SMHIP_DEV=1 SMHIP_TRACE_ALLOC=1 SIGMAP_STREAM_STATS=2 timeout 300 sigmap_amd/bin/sigmap -m -r $W/ref.fa -p $W/model.txt -x $W/idx -s $W/reads_r10000_c8 -o $W/out2.paf --streams 2 > $O/trace.out 2> $O/trace.err
grep -v "hipMalloc 0.00" $O/trace.err | cut -c1-200 | head -150
