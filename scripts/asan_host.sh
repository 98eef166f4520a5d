#!/bin/bash
# The host runtime (gvs_host.cpp, gvk_host.cpp) under AddressSanitizer + UBSan: built host-only with g++ (the device entry
# points are stubs that return GVK_EHIP) and run through the host test suites, fat and thin table forms, AVX-512 and AVX2
# uniform generators.  No GPU needed.  Usage: bash scripts/asan_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/gvk_asan
mkdir -p $OUT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread \
    -I$ROOT/include -I$ROOT/graphvite_amd/csrc -I/opt/rocm/include $ROOT/graphvite_amd/csrc/gvs_host.cpp \
    $ROOT/graphvite_amd/csrc/gvk_host.cpp $ROOT/scripts/asan/device_stubs.cpp -L/opt/rocm/lib -lroctx64 \
    -Wl,-rpath,/opt/rocm/lib -o $OUT/libgvk.so
GCC_LIB=$(dirname $(g++ -print-file-name=libasan.so))
run() {
  LD_PRELOAD=$GCC_LIB/libasan.so:$GCC_LIB/libubsan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -c "
import graphvite_amd._lib as L
L.LIB_PATH='$OUT/libgvk.so'
import pytest, sys
sys.exit(pytest.main(['tests/test_host_cpu.py', 'tests/test_solver_cpu.py', '-q', '-p', 'no:cacheprovider', '-k',
                      'not gloo and not simd and not thin_tables and not dry_run and not error_codes and not exports_every']))"
}
cd $ROOT
run
GVS_FAT_SLOT_LIMIT=0 GVS_NO_AVX512=1 run
