#!/bin/bash
# Measurement build: folve_amd/variants/libfolve_amd_p16.so = the library with -DFOLVE_EXPERIMENT_P16 (kernels instantiated at
# P = 16384 too; FOLVE_P16=1 in the environment makes every filter longer than 16 384 taps use that partition).  Then, on
# the GPU box:   python tools/p16_experiment.py; FOLVE_P16=1 FOLVE_AMD_LIB=$PWD/folve_amd/variants/libfolve_amd_p16.so python tools/p16_experiment.py
set -e
cd "$(dirname "$0")/.."
make -s -C folve_amd/csrc >/dev/null
mkdir -p folve_amd/variants /tmp/fk_p16
F="-O3 -std=c++20 -fPIC -DFOLVE_EXPERIMENT_P16"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c folve_amd/csrc/kernels/kernels.hip -o /tmp/fk_p16/kernels.o
/opt/rocm/bin/hipcc $F -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c folve_amd/csrc/engine.cpp -o /tmp/fk_p16/engine.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o folve_amd/variants/libfolve_amd_p16.so /tmp/fk_p16/kernels.o /tmp/fk_p16/engine.o \
    $(find folve_amd/csrc/build/host -name '*.o') -lpthread
echo built folve_amd/variants/libfolve_amd_p16.so
