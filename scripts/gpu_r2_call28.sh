#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 200 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn or attention" 2>&1 | tail -3
for lib in scratch_libs/lib_base.so scratch_libs/lib_attn3.so; do
  echo "## $lib"; MICRODIT_LIB=$lib timeout -k 5 150 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep "fwd\|split"
done | tee gpurun_out/c28_attn.log
