#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (host library, oracle) through the CPU test suite.
# (GPU sanitizers are not available on this pool; the device code is covered by the parity tests.)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/sigmap_asan}
mkdir -p $OUT
H=sigmap_amd/host
# (built without HDF5: an rpath to /opt/conda/lib would also resolve libasan to conda's older copy and two
# ASan runtimes cannot share a process; the FAST5 reader is therefore not covered by this run)
g++ -std=c++17 -O1 -g -ffp-contract=off -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $OUT/libsigmap_host.so \
    $H/kdtree.cc $H/pore_model.cc $H/sequence_batch.cc $H/signal_batch.cc $H/fast5_reader.cc $H/spatial_index.cc \
    $H/mapping_logic.cc $H/host_abi.cc -lz
gcc -std=c99 -O1 -g -ffp-contract=off -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $OUT/libsigmap_oracle.so oracle/sigmap_oracle.c -lm
export SMHOST_LIB=$OUT/libsigmap_host.so SM_ORACLE_LIB=$OUT/libsigmap_oracle.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export PYTEST_DISABLE_PLUGIN_AUTOLOAD=1  # (third-party pytest plugins pull in extension modules that abort under a preloaded ASan)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
# (the files that do not import torch: its runtime does not start under a preloaded ASan)
python -m pytest tests/test_oracle_golden.py tests/test_host_index.py tests/test_fast5.py tests/test_dtw_path.py -x -q -m "not gpu" "${@:2}"
