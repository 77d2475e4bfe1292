#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_variants_gpu.py -q -k "attention or race or plain" 2>&1 | tail -15 ) > gpurun_out/c6_pytest.log 2>&1
( timeout 600 python scripts/bench_attn.py 10 1024 2>&1 | tail -20 ) > gpurun_out/c6_attn.log 2>&1
( timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_trainer_gpu.py -q 2>&1 | tail -8 ) > gpurun_out/c6_pytest_engine.log 2>&1
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages 2>&1 | tail -2 ) > gpurun_out/c6_bench.log 2>&1
tail -5 gpurun_out/c6_pytest.log; cat gpurun_out/c6_attn.log; tail -4 gpurun_out/c6_pytest_engine.log; cat gpurun_out/c6_bench.log
