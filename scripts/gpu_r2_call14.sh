#!/bin/bash
# pp256 epilogue v2 (single predicated quadrant, store-aware vmcnt counts): parity, A/B against the previous library, timeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/c14_pytest.log
tail -4 gpurun_out/c14_pytest.log
if ! grep -q " passed" gpurun_out/c14_pytest.log || grep -q "failed" gpurun_out/c14_pytest.log; then echo "PARITY FAILED - stopping"; exit 1; fi
for lib in scratch_libs/lib_base.so scratch_libs/lib_epi2.so; do
  echo "## $lib"
  MICRODIT_LIB=$lib timeout -k 5 200 python scripts/bench_gemm_variants.py --variants pp256 --rounds 2 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c14_ab.log 2>&1
paste <(grep -v "^#" gpurun_out/c14_ab.log | head -25) <(grep -v "^#" gpurun_out/c14_ab.log | tail -25 | awk '{print $NF}')
{
timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16
timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 res
timeout -k 5 100 python scripts/gemm_pp_timeline.py 16384 3840 1024 1 0 gelu 8
timeout -k 5 100 python scripts/gemm_pp_timeline.py 1024 1024 65536 0 0 f32 1 16
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c14_timeline.log
grep "shape\|tile 0\|tile 1\|last epi" gpurun_out/c14_timeline.log
