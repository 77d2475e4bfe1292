#!/bin/bash
# step-level A/B of the attention (batch, head) -> XCD mapping: separate processes, A B A B, same box; attention class time from the profile leg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
  for lib in noremap remap; do
      v=$(MICRODIT_LIB=scratch_libs/lib_$lib.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-stages 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); a=d['roofline_hbm']['kernels']['attention']; print('%.1f images/s  %.1f ms; attention %.2f ms in %d launches, %.3f of HBM peak' % (d['value'], d['ms_per_step'], a['total_ms'], a['launches'], a['frac_of_hbm_peak']))")
      echo "$lib round $round: $v"
  done
done | tee gpurun_out/x5_attn_xcd_step_ab.txt
MICRODIT_LIB=scratch_libs/lib_remap.so python -c "
from micro_diffusion_amd import hip; import ctypes; print(hip.lib()._name)"
