#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_gpu.py -q -x 2>&1 | tail -12 > gpurun_out/c26_pytest.log; tail -4 gpurun_out/c26_pytest.log
if grep -q "failed" gpurun_out/c26_pytest.log; then grep -B12 "Error" gpurun_out/c26_pytest.log | head -50; fi
libs="scratch_libs/lib_base.so scratch_libs/lib_quadrows.so"
for r in 1 2; do for lib in $libs; do
  MICRODIT_LIB=$lib timeout -k 5 120 python scripts/bench_gemm_variants.py --variants pp256 --rounds 2 2>&1 | grep -v "amdgpu.ids\|^#" > gpurun_out/c26_$(basename $lib .so)_$r.txt
done; done
cd gpurun_out
paste <(cat c26_lib_base_1.txt) <(awk '{print $NF}' c26_lib_quadrows_1.txt) <(awk '{print $NF}' c26_lib_base_2.txt) <(awk '{print $NF}' c26_lib_quadrows_2.txt) | tee c26_ab.log
cd ..
MICRODIT_LIB=scratch_libs/lib_quadrows.so timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16 2>&1 | grep -v amdgpu.ids | grep "shape\|tile 0\|tile 1\|last epi\|exit skew" | tee gpurun_out/c26_timeline.log
MICRODIT_LIB=scratch_libs/lib_quadrows.so timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 res 2>&1 | grep -v amdgpu.ids | grep "shape\|tile 0\|tile 1\|last epi\|exit skew" | tee -a gpurun_out/c26_timeline.log
