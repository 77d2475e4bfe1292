#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_gemm_variants_gpu.py tests/test_gemm_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -6 > gpurun_out/c27_pytest.log; tail -4 gpurun_out/c27_pytest.log
for lib in scratch_libs/lib_base.so scratch_libs/lib_quadrows.so scratch_libs/lib_base.so scratch_libs/lib_quadrows.so; do
  echo "## $lib"; MICRODIT_LIB=$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4))"
done | tee gpurun_out/c27_bench_ab.log
