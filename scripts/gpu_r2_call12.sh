#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in scratch_libs/libpp_*.so; do
  echo "## $lib"
  MICRODIT_LIB=$lib timeout 120 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c12_epi_bisect.log 2>&1
cat gpurun_out/c12_epi_bisect.log
