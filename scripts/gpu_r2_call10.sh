#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in scratch_libs/libpph_*.so; do
  echo "## $lib"
  MICRODIT_LIB=$lib timeout 200 python scripts/bench_gemm_variants.py --variants pp256,pph256 2>&1 | grep "bb proj/q fwd\|8k cube\|moe fc1 fwd f\|bb dgrad 1024\|wgrad 1024x"
done > gpurun_out/c10_bisect.log 2>&1
cat gpurun_out/c10_bisect.log
