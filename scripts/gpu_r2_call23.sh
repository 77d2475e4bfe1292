#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
libs="scratch_libs/lib_plain.so scratch_libs/lib_nt.so"
for r in 1 2; do for lib in $libs; do
  MICRODIT_LIB=$lib timeout -k 5 120 python scripts/bench_gemm_variants.py --variants pp256 --rounds 2 2>&1 | grep -v "amdgpu.ids\|^#" > gpurun_out/c23_$(basename $lib .so)_$r.txt
done; done
cd gpurun_out
paste <(cat c23_lib_plain_1.txt) <(awk '{print $NF}' c23_lib_nt_1.txt) <(awk '{print $NF}' c23_lib_plain_2.txt) <(awk '{print $NF}' c23_lib_nt_2.txt) | tee c23_ab.log
cd ..
MICRODIT_LIB=scratch_libs/lib_nt.so timeout -k 5 100 python scripts/gemm_pp_timeline.py 65536 1024 1024 1 1 bf16 2>&1 | grep -v amdgpu.ids | grep "shape\|tile 0\|tile 1\|last epi\|exit skew" | tee gpurun_out/c23_timeline.log
