#!/bin/bash
# which of the v2 epilogue changes costs the k-loop: C = 0 MFMA branch / store-aware vmcnt
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
libs="scratch_libs/lib_base.so scratch_libs/lib_epi2.so scratch_libs/lib_zeroepi.so scratch_libs/lib_flatvm.so scratch_libs/lib_zeroflat.so"
for lib in $libs; do
  MICRODIT_LIB=$lib timeout -k 5 120 python scripts/bench_gemm_variants.py --variants pp256 --rounds 2 2>&1 | grep -v "amdgpu.ids\|^#" > gpurun_out/c15_$(basename $lib .so).txt
done
cd gpurun_out
echo "$libs" | tr ' ' '\n' | xargs -n1 basename | tr '\n' ' '; echo
paste <(cat c15_lib_base.txt) <(awk '{print $NF}' c15_lib_epi2.txt) <(awk '{print $NF}' c15_lib_zeroepi.txt) <(awk '{print $NF}' c15_lib_flatvm.txt) <(awk '{print $NF}' c15_lib_zeroflat.txt) | tee c15_ab.log
