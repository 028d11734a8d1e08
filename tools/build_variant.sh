#!/bin/bash
# Tuning build: libmesm with extra -D flags on ONE source (SRC=gemm.hip by default, e.g. SRC=attention.hip)
# -> mesm_amd/variants/libmesm_<name>.so (git-ignored, travels with gpurun).
# Use with MESM_LIB_PATH=mesm_amd/variants/libmesm_<name>.so
name=$1; shift
src=${SRC:-gemm.hip}
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/mesm_amd/variants
objs=""
for f in $root/mesm_amd/csrc/build/*.hip.o; do case $f in *$src.o) ;; *) objs="$objs $f";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-gpu-rdc -I $root/include -I $root/mesm_amd/csrc -Wno-unused-value "$@" -c $root/mesm_amd/csrc/$src -o /tmp/var_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/mesm_amd/variants/libmesm_$name.so /tmp/var_$name.o $objs && echo built $name
