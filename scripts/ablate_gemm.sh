#!/bin/bash
# timing ablations of the 2-stage LDS-DMA GEMM (results invalid when flags are set)
for v in dma256 dma128; do
for f in 0 1 2 3 4 7 8 15; do
  echo "variant $v  MD_GEMM_DEBUG=$f  (1=noDMA 2=noLDSread 4=noBarrier 8=noEpilogue)"
  MD_GEMM_VARIANT=$v MD_GEMM_DEBUG=$f python scripts/bench_gemm.py 2>&1 | grep -E "bb qkv fwd|8k cube|bb proj"
done; done
