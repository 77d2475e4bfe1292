#!/bin/bash
# Same-box A/B of libmicrodit_hip.so builds.  Boxes of the pool differ by +-3-5 %, and a forced variant name that the
# library does not know silently falls back to its own choice, so kernel experiments are compared like this:
#
#   1. build candidate libraries here (hipcc cross-compiles without a GPU) and park them under scratch_libs/ (git-ignored,
#      but shipped to the GPU box):   python -c "from micro_diffusion_amd import hip; hip.build()" && cp micro_diffusion_amd/libmicrodit_hip.so scratch_libs/libB.so
#   2. run them back to back in ONE gpurun call:   gpurun -- 'bash scripts/ab_libs.sh 1024 scratch_libs/libA.so scratch_libs/libB.so'
#
# Prints the per-class GEMM summary of scripts/profile_gemms.py for every library, twice (A B A B) to expose drift.
mb=${1:-1024}; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "## $lib (round $round, microbatch $mb)"
    MICRODIT_LIB=$lib timeout 120 python scripts/profile_gemms.py "$mb" 2>&1 | grep "microbatch\|fwd\|dgrad\|wgrad"
  done
done
