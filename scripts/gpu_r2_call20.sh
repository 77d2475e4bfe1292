#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in scratch_libs/lib_plain.so scratch_libs/lib_quad.so scratch_libs/lib_line.so scratch_libs/lib_nostore.so; do
  echo "## $lib"
  MICRODIT_LIB=$lib timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16 2>&1 | grep -v amdgpu.ids | grep "shape\|tile 0\|tile 1\|tile 2\|last epi\|exit skew"
done | tee gpurun_out/c20_store_pattern.log
