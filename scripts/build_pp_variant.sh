#!/bin/bash
# Build scratch_libs/lib_<name>.so = the current library with ONE source recompiled under extra -D flags (kernel experiments; the
# other objects are taken from csrc/build as they are).  Usage: scripts/build_pp_variant.sh <name> [--src attention.hip] [-DX_... ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=gemm_pp.hip
if [ "$1" == "--src" ]; then src=$2; shift 2; fi
base=${src%.hip}
mkdir -p scratch_libs/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result "$@" -I include -c micro_diffusion_amd/csrc/$src -o scratch_libs/obj/${base}_$name.o
objs=$(ls micro_diffusion_amd/csrc/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scratch_libs/obj/${base}_$name.o -o scratch_libs/lib_$name.so
echo built scratch_libs/lib_$name.so
