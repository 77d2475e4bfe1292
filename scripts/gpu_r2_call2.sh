#!/bin/bash
# round 2, GPU call 2: split-K factors for pp256, full GPU suite with pp256 in the AUTO rules, end-to-end bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 300 python scripts/bench_wgrad.py --variant pp256 2>&1 | tail -20 ) > gpurun_out/c2_wgrad_pp256_mb1024.log 2>&1
( timeout 300 python scripts/bench_wgrad.py --variant pp256 --mb 256 2>&1 | tail -20 ) > gpurun_out/c2_wgrad_pp256_mb256.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > gpurun_out/c2_pytest.log 2>&1
( timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | tail -3 ) > gpurun_out/c2_bench.log 2>&1
( timeout 600 python scripts/profile_gemms.py 1024 2>&1 | tail -60 ) > gpurun_out/c2_profile_gemms_1024.log 2>&1
( timeout 600 python scripts/bench_gemm_variants.py --mb 256 2>&1 | tail -40 ) > gpurun_out/c2_bench256.log 2>&1
tail -12 gpurun_out/c2_pytest.log; cat gpurun_out/c2_bench.log; cat gpurun_out/c2_wgrad_pp256_mb1024.log
