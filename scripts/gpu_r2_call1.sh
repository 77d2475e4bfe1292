#!/bin/bash
# round 2, GPU call 1: the new GEMM kernel — parity (all variants, real shapes), race screen, per-shape throughput
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gemm_variants_gpu.py -x -q -k "race_screen or refuses" 2>&1 | tail -15 ) > gpurun_out/c1_sanity.log 2>&1
( timeout 900 python -m pytest tests/test_gemm_variants_gpu.py -q 2>&1 | tail -40 ) > gpurun_out/c1_variants.log 2>&1
( timeout 600 python scripts/bench_gemm_variants.py --mb 1024 2>&1 | tail -40 ) > gpurun_out/c1_bench1024.log 2>&1
tail -5 gpurun_out/c1_sanity.log; tail -8 gpurun_out/c1_variants.log; cat gpurun_out/c1_bench1024.log
