#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 200 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln or qkln or adamw or edm or moe" 2>&1 | tail -4
for lib in scratch_libs/lib_base.so scratch_libs/lib_norm2.so scratch_libs/lib_base.so scratch_libs/lib_norm2.so; do
  echo "## $lib"; MICRODIT_LIB=$lib timeout -k 5 100 python scripts/bench_norm.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/c17_norm.log
