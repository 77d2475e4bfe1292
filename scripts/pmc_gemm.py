"""One GEMM shape, one kernel variant, N launches — the workload for rocprofv3 --pmc passes on the GEMM kernels.
    python scripts/pmc_gemm.py M N K akc bkc variant [mode b|r|g] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from micro_diffusion_amd import hip  # noqa: E402

M, N, K, akc, bkc = [int(v) for v in sys.argv[1:6]]
variant = sys.argv[6]
mode = sys.argv[7] if len(sys.argv) > 7 else "b"
n = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = "cuda"
A = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
B = (torch.randn((N, K) if bkc else (K, N), device=dev) * 0.05).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
kw = dict(lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc, variant=hip.GEMM_VARIANT_NAMES[variant])
if mode == "r":
    res = torch.randn(M, N, device=dev).bfloat16()
    gate = torch.randn(M // 64, N, device=dev).bfloat16()
    c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw.update(mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=64, C2=c2, ldc2=N)
elif mode == "g":
    c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw.update(act=hip.ACT_GELU_ERF, C2=c2, ldc2=N)
for _ in range(n):
    hip.gemm(A, B, C, M, N, K, **kw)
torch.cuda.synchronize()
print("done", M, N, K, variant, mode)
