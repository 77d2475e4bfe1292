#!/bin/bash
# bench + kernel-trace profile of the step with the v2 pp256 epilogue
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages > gpurun_out/c16_bench.log 2>&1
tail -1 gpurun_out/c16_bench.log | cut -c1-700
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_c16 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile > $GRAFT_REPO_ROOT/gpurun_out/c16_rocprof.log 2>&1 )
find gpurun_out/prof_r2_c16 -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
tail -1 gpurun_out/c16_rocprof.log | cut -c1-300
ls -R gpurun_out/prof_r2_c16 | head
