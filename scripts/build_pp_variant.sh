#!/bin/bash
# Build scratch_libs/lib_<name>.so = the current library with gemm_pp.hip recompiled under extra -D flags (kernel experiments;
# the other objects are taken from csrc/build as they are).  Usage: scripts/build_pp_variant.sh <name> [-DPP_X_... ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scratch_libs/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -I include -c micro_diffusion_amd/csrc/gemm_pp.hip -o scratch_libs/obj/gemm_pp_$name.o
objs=$(ls micro_diffusion_amd/csrc/build/*.o | grep -v gemm_pp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scratch_libs/obj/gemm_pp_$name.o -o scratch_libs/lib_$name.so
echo built scratch_libs/lib_$name.so
