cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=${1:-scratch_libs/lib_w4_dma.so}
MICRODIT_LIB=$L timeout 600 python -m pytest tests/test_gemm_variants_gpu.py -m gpu -q -x -k "w4 or swiglu" 2>&1 | tail -5
bash scripts/ab_w4.sh w4,pp256 scratch_libs/lib_w4_base.so $L 2>&1 | tee gpurun_out/w4_dma_ab_nt.txt
W4_BKC=0 bash scripts/ab_w4.sh w4,pp256 scratch_libs/lib_w4_base.so $L 2>&1 | tee gpurun_out/w4_dma_ab_nn.txt
