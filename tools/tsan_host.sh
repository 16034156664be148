#!/bin/bash
# ThreadSanitizer build of the HOST side of libfolve_amd.so, run against the combiner's CPU stress test
# (16 threads through folve::BatchScheduler).  Device code is unchanged; no GPU needed.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/folve_tsan; mkdir -p $B
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O1 -g -std=c++20 -fPIC -fsanitize=thread -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for f in $R/folve_amd/csrc/engine.cpp $R/folve_amd/csrc/trace.cpp $R/folve_amd/csrc/host/*.cpp; do
  $CLANG $FLAGS -c $f -o $B/$(basename $f).o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -o $B/libfolve_amd_tsan.so $B/*.o $R/folve_amd/csrc/build/kernels/kernels.o $R/folve_amd/csrc/build/kernels/mac_walk3.o -lpthread -ldl
RT=$($CLANG -print-file-name=libclang_rt.tsan-x86_64.so)
cd $R
TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 LD_PRELOAD=$RT FOLVE_AMD_LIB=$B/libfolve_amd_tsan.so \
  python -m pytest tests/test_host_cpu.py -x -q -p no:cacheprovider -k "(combiner or pool) and not sanitizer" "$@"
