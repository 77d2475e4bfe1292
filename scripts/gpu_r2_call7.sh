#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_data_gpu.py tests/test_gemm_gpu.py -q 2>&1 | tail -8 ) > gpurun_out/c7_pytest.log 2>&1
( timeout 600 python scripts/profile_gemms.py 256 2>&1 | tail -60 ) > gpurun_out/c7_profile_gemms_256.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_mb256 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --microbatch 256 --no-cpu-baseline --no-other-stages --no-profile 2>&1 | tail -2 ) > gpurun_out/c7_rocprof_mb256.log 2>&1
find gpurun_out/prof_r2_mb256 -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
tail -4 gpurun_out/c7_pytest.log; tail -3 gpurun_out/c7_rocprof_mb256.log; head -60 gpurun_out/c7_profile_gemms_256.log
