#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_kernels_gpu.py tests/test_trainer_gpu.py -q -k "two_ranks or adamw or three_steps" 2>&1 | tail -25 ) > gpurun_out/c4_pytest.log 2>&1
( timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -3 ) > gpurun_out/c4_bench.log 2>&1
( MD_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --microbatch 256 --no-cpu-baseline 2>&1 | tail -5 ) > gpurun_out/c4_bench_dp2_gloo.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2a -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages 2>&1 | tail -3 ) > gpurun_out/c4_rocprof.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python $GRAFT_REPO_ROOT/scripts/pmc_workload.py 2>&1 | tail -3 ) > gpurun_out/c4_pmc_fetch.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python $GRAFT_REPO_ROOT/scripts/pmc_workload.py 2>&1 | tail -3 ) > gpurun_out/c4_pmc_write.log 2>&1
( python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r2_gemm_traffic.json 2>&1 | tail -40 ) > gpurun_out/c4_traffic.log 2>&1
# keep only the small summaries of the profiler output
find gpurun_out/prof_r2a gpurun_out/pmc_fetch gpurun_out/pmc_write -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
tail -8 gpurun_out/c4_pytest.log; cat gpurun_out/c4_bench.log; tail -3 gpurun_out/c4_bench_dp2_gloo.log; tail -3 gpurun_out/c4_rocprof.log; tail -5 gpurun_out/c4_traffic.log
