#!/bin/bash
# Phase breakdown of an instrumented kernel (PHASE_MARK in sort.hip).  Two steps:
#   tools/phase_prof.sh build            (here: a second library with -DSMHIP_PHASE_PROF, sigmap_amd/lib/libsigmap_hip_prof.so)
#   tools/phase_prof.sh run <bench args> (GPU box: bench.py's serial pass on that library; the shares go to stderr)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift
  EXTRA="$*"   # further -D flags (experiments)
  mkdir -p build/prof
  for f in api signal events seed_lookup sort chain sdtw cwt; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -DSMHIP_PHASE_PROF $EXTRA \
      -c sigmap_amd/csrc/$f.hip -o build/prof/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sigmap_amd/lib/libsigmap_hip_prof.so build/prof/*.o -ldl
  echo built sigmap_amd/lib/libsigmap_hip_prof.so
else
  shift
  SMHIP_LIB=$PWD/sigmap_amd/lib/libsigmap_hip_prof.so python bench.py --steps 3 --warmup 1 --no-cpu --no-pcie --depth 1 "$@" 2>&1 | grep -E "phase_prof|walk_prof"
fi
