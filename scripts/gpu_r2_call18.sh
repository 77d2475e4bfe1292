#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_gpu.py -q -x 2>&1 | tail -4
libs="scratch_libs/lib_base.so scratch_libs/lib_plain.so"
for r in 1 2; do for lib in $libs; do
  MICRODIT_LIB=$lib timeout -k 5 120 python scripts/bench_gemm_variants.py --variants pp256 --rounds 2 2>&1 | grep -v "amdgpu.ids\|^#" > gpurun_out/c18_$(basename $lib .so)_$r.txt
done; done
cd gpurun_out
paste <(cat c18_lib_base_1.txt) <(awk '{print $NF}' c18_lib_plain_1.txt) <(awk '{print $NF}' c18_lib_base_2.txt) <(awk '{print $NF}' c18_lib_plain_2.txt) | tee c18_ab.log
