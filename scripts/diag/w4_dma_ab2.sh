cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for L in "$@"; do MICRODIT_LIB=$L timeout 600 python -m pytest tests/test_gemm_variants_gpu.py -m gpu -q -x -k "w4 or swiglu" 2>&1 | tail -1; done
bash scripts/ab_w4.sh w4 scratch_libs/lib_w4_base.so "$@" 2>&1 | tee gpurun_out/w4_dma_ab2_nt.txt
W4_BKC=0 bash scripts/ab_w4.sh w4 scratch_libs/lib_w4_base.so "$@" 2>&1 | tee gpurun_out/w4_dma_ab2_nn.txt
