"""One GEMM shape, many launches (for rocprofv3 --pmc): python scripts/bench_gemm_one.py M N K akc bkc [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd import hip
M, N, K, akc, bkc = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
A = torch.randn((M, K) if akc else (K, M), device="cuda").bfloat16()
B = torch.randn((N, K) if bkc else (K, N), device="cuda").bfloat16()
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    hip.gemm(A, B, C, M, N, K, lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc)
torch.cuda.synchronize()
