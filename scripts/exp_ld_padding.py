"""Does a power-of-two leading dimension cost the K-contiguous operand loads (channel camping)?  One NT / NN shape, the operands'
leading dimensions padded by `pad` elements, w4 and pp256.  python scripts/exp_ld_padding.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd import hip


def run(M, N, K, akc, bkc, pad_a, pad_b, pad_c, variant, iters=10):
    lda = (K if akc else M) + pad_a
    ldb = (K if bkc else N) + pad_b
    ldc = N + pad_c
    A = torch.randn((M if akc else K), lda, device="cuda").bfloat16()
    B = torch.randn((N if bkc else K), ldb, device="cuda").bfloat16()
    C = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hip.gemm(A, B, C, M, N, K, lda=lda, ldb=ldb, ldc=ldc, a_kcontig=akc, b_kcontig=bkc, variant=variant)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            hip.gemm(A, B, C, M, N, K, lda=lda, ldb=ldb, ldc=ldc, a_kcontig=akc, b_kcontig=bkc, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return 2.0 * M * N * K / best / 1e9


shapes = [(65536, 1024, 1024, 1, 1), (65536, 1024, 1024, 1, 0), (65536, 3072, 1024, 1, 1), (65536, 1024, 3072, 1, 0), (262144, 768, 768, 1, 1),
          (65536, 5632, 1024, 1, 1), (16384, 1024, 1024, 1, 1)]
print("shape                         kc  variant |   pad 0/0/0   pad A64   pad B64   pad AB64  pad ABC64  pad AB 8   pad AB 192")
for M, N, K, akc, bkc in shapes:
    for vname, v in (("w4", hip.GEMM_W4), ("pp256", hip.GEMM_PP256)):
        row = []
        for pa, pb, pc in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (8, 8, 0), (192, 192, 0)):
            row.append(run(M, N, K, akc, bkc, pa, pb, pc, v))
        print(f"{M:7d}x{N:5d}x{K:5d}          {akc}{bkc}  {vname:6s}  | " + "  ".join(f"{r:8.0f}" for r in row), flush=True)
