#!/bin/bash
# A/B of the (batch, head) -> XCD mapping of the attention kernels (scratch_libs/lib_plain.so vs lib_xcd.so), A B A B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
  for lib in plain xcd; do
    echo "## $lib (round $round)"
    MICRODIT_LIB=scratch_libs/lib_$lib.so timeout 200 python scripts/bench_attn.py 20 1024 2>&1 | grep "fwd\|bwd auto"
  done
done > gpurun_out/x3_attn_xcd_ab.txt 2>&1
MICRODIT_LIB=scratch_libs/lib_xcd.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k attention 2>&1 | tail -2
cat gpurun_out/x3_attn_xcd_ab.txt
