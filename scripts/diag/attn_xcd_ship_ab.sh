#!/bin/bash
# the shipped window-of-8-samples mapping against the sweep library at G = 16 and at contiguous eighths (A B C A B C)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
  for v in "xcdg 0" "xcdg 16" "ship 0"; do
    set -- $v
    echo "## lib_$1 G=$2 (round $round)"
    MD_ATTN_XCD_G=$2 MICRODIT_LIB=scratch_libs/lib_$1.so timeout 200 python scripts/bench_attn.py 20 1024 2>&1 | grep "fwd\|bwd auto" | grep -v "S=1024\|1024x77\|S=256\|256x77"
  done
done 2>&1 | tee gpurun_out/x8_attn_xcd_ship_ab.txt
MICRODIT_LIB=scratch_libs/lib_ship.so timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k attention 2>&1 | tail -2
