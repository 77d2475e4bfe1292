"""Diagnostic for the -0.47 % loss offset at XL/2 (profiles/parity): the residual stream keeps gain 1.000 through blocks.27 but the
network output has gain 0.9953 against the oracle, so the offset is made in the final layer.  Each op of the final layer
(utils.py:236-240) against torch fp32 on the same bf16-rounded inputs, with the least-squares gain next to the rel-RMS."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ctypes import byref  # noqa: E402

from micro_diffusion_amd import hip  # noqa: E402

dev = "cuda"
L, st = hip.lib(), hip.stream_ptr()


def gain(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (b * b).sum())


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


torch.manual_seed(0)
for M, N, K in [(128, 16, 1024), (256, 16, 1024), (512, 16, 1024), (128, 16, 256), (2, 2048, 1024), (2, 6144, 1024), (128, 1024, 1024)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ch = []
    hip.gemm(A=A, B=W, C=C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=b, chosen=ch)
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + b
    print(f"gemm {M}x{N}x{K} variant {ch}: gain {gain(C.float(), ref):.5f} rel {rel(C.float(), ref):.5f}")
# modulated LayerNorm, rows_per_sample 64 / 256
for B, S, Cc in [(2, 64, 1024), (1, 256, 1024), (4, 64, 256)]:
    rows = B * S
    x = torch.randn(rows, Cc, device=dev).bfloat16()
    w = torch.randn(Cc, device=dev) * 0.1 + 1
    mod = torch.randn(B, 2 * Cc, device=dev).bfloat16()
    out = torch.empty(rows, Cc, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    a = hip.LnArgs(x.data_ptr(), w.data_ptr(), mod.data_ptr(), mod.data_ptr() + 2 * Cc, None, out.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   rows, Cc, Cc, Cc, 2 * Cc, S, 0, 1e-6, 0)
    hip.check(L.md_ln_fwd(byref(a), st), "ln")
    torch.cuda.synchronize()
    xn = torch.nn.functional.layer_norm(x.float(), (Cc,), w, None, 1e-6).view(B, S, Cc)
    ref = (xn * (1 + mod[:, Cc:].float().unsqueeze(1)) + mod[:, :Cc].float().unsqueeze(1)).view(rows, Cc)
    print(f"modulated LN B={B} S={S} C={Cc}: gain {gain(out.float(), ref):.5f} rel {rel(out.float(), ref):.5f}")
