#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/c8_pytest.log 2>&1
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/c8_bench.log 2>&1
tail -6 gpurun_out/c8_pytest.log; cat gpurun_out/c8_bench.log
