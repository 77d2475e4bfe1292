#!/bin/bash
# Build scratch_libs/lib_w4_<name>.so = the current library with gemm_w4.hip recompiled on a k-loop schedule generated with the given
# schedule flags (scripts/gen_w4_acc.py: c1=<n> c2=<n>).  Usage: scripts/build_w4_variant.sh <name> [flag ...] [-- -DX ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
flags=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do flags+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p scratch_libs/obj
python scripts/gen_w4_acc.py "$PWD/scratch_libs/obj/w4_$name.inc" "${flags[@]}" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-inline-asm "$@" \
  "-DW4_ACC_INC=\"$PWD/scratch_libs/obj/w4_$name.inc\"" -I include -c micro_diffusion_amd/csrc/gemm_w4.hip -o scratch_libs/obj/gemm_w4_$name.o
objs=$(ls micro_diffusion_amd/csrc/build/*.o | grep -v "/gemm_w4.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scratch_libs/obj/gemm_w4_$name.o -o scratch_libs/lib_w4_$name.so
echo built scratch_libs/lib_w4_$name.so
