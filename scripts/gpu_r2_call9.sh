#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_variants_gpu.py -q -x -k "pph256" 2>&1 | tail -15 ) > gpurun_out/c9_pytest.log 2>&1
( timeout 600 python scripts/bench_gemm_variants.py --variants pp256,pph256 2>&1 | tail -30 ) > gpurun_out/c9_variants.log 2>&1
tail -8 gpurun_out/c9_pytest.log; cat gpurun_out/c9_variants.log
