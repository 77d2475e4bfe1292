#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 120 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16
timeout 120 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 res
timeout 120 python scripts/gemm_pp_timeline.py 65536 3072 1024 1 1 bf16
timeout 120 python scripts/gemm_pp_timeline.py 16384 3840 1024 1 0 gelu 8
timeout 120 python scripts/gemm_pp_timeline.py 8192 8192 8192 1 1 bf16
timeout 120 python scripts/gemm_pp_timeline.py 1024 1024 65536 0 0 f32 1 16
} > gpurun_out/c11_timeline.log 2>&1
cat gpurun_out/c11_timeline.log
