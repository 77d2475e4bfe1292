#!/bin/bash
# A/B of the workgroup -> XCD mapping of the LayerNorm family (scratch_libs/lib_plain.so vs lib_xcd.so), A B A B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
  for lib in plain xcd; do
    echo "## $lib (round $round)"
    MICRODIT_LIB=scratch_libs/lib_$lib.so timeout 200 python scripts/bench_norm.py --iters 20 2>&1 | grep "TB/s"
  done
done > gpurun_out/x2_norm_xcd_ab.txt 2>&1
cat gpurun_out/x2_norm_xcd_ab.txt
MICRODIT_LIB=scratch_libs/lib_xcd.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -2
