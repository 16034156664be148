#!/bin/bash
# AddressSanitizer + UBSan build of the HOST side of libfolve_amd.so (engine, loader, pool, batcher),
# run against the CPU test-suite.  GPU ASan is not available on this pool; device code is unchanged.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/folve_asan; mkdir -p $B
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O1 -g -std=c++20 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for f in $R/folve_amd/csrc/engine.cpp $R/folve_amd/csrc/trace.cpp $R/folve_amd/csrc/host/*.cpp; do
  $CLANG $FLAGS -c $f -o $B/$(basename $f).o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $B/libfolve_amd_asan.so $B/*.o $R/folve_amd/csrc/build/kernels/kernels.o $R/folve_amd/csrc/build/kernels/mac_walk3.o -lpthread -ldl
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 LD_PRELOAD=$RT FOLVE_AMD_LIB=$B/libfolve_amd_asan.so \
  python -m pytest tests/test_host_cpu.py -x -q -p no:cacheprovider -k "not sanitizer" "$@"   # (the tests that build their own sanitizer binaries run in the plain suite)
