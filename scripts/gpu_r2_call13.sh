#!/bin/bash
# fast pp256 epilogue: parity of every GEMM variant, A/B against the previous library, timeline, step profile
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_gpu.py -q -x 2>&1 | tail -15 ) > gpurun_out/c13_pytest.log 2>&1
tail -5 gpurun_out/c13_pytest.log
for lib in scratch_libs/lib_base.so scratch_libs/lib_fastepi.so; do
  echo "## $lib"
  MICRODIT_LIB=$lib timeout 300 python scripts/bench_gemm_variants.py --variants pp256 --rounds 3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c13_ab.log 2>&1
paste <(grep -v "^#" gpurun_out/c13_ab.log | head -25) <(grep -v "^#" gpurun_out/c13_ab.log | tail -25 | awk '{print $NF}')
{
timeout 120 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16
timeout 120 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 res
timeout 120 python scripts/gemm_pp_timeline.py 16384 3840 1024 1 0 gelu 8
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c13_timeline.log
grep "shape\|tile 0\|tile 1\|last epi" gpurun_out/c13_timeline.log
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_c13 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages 2>&1 | tail -2 ) > gpurun_out/c13_rocprof.log 2>&1
find gpurun_out/prof_r2_c13 -type f ! -name "*stats*" -size +2M -delete 2>/dev/null
tail -1 gpurun_out/c13_rocprof.log | cut -c1-400
