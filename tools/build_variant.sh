#!/bin/bash
# Dev aid: build folve_amd/variants/libfolve_amd_<name>.so with extra -D flags on the kernels
# (host objects are reused from the product build).  usage: tools/build_variant.sh name -DFOO -DBAR
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C folve_amd/csrc >/dev/null
mkdir -p folve_amd/variants /tmp/fkv_$name
for k in kernels mac_walk3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC "$@" -c folve_amd/csrc/kernels/$k.hip -o /tmp/fkv_$name/$k.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o folve_amd/variants/libfolve_amd_$name.so /tmp/fkv_$name/kernels.o /tmp/fkv_$name/mac_walk3.o \
    $(find folve_amd/csrc/build -name '*.o' ! -name kernels.o ! -name mac_walk3.o) -lpthread -ldl
echo built folve_amd/variants/libfolve_amd_$name.so
