"""Where the w4 gated-residual raw copy differs from the reference (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from micro_diffusion_amd import hip
dev = "cuda"
M, N, K, rps = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (65536, 1024, 2816, 64)
torch.manual_seed(7 + M + K)
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
res = torch.randn(M, N, device=dev).to(torch.bfloat16)
gate = torch.randn((M + rps - 1) // rps, N, device=dev).to(torch.bfloat16)
raw = A.float() @ B.float().t()
for variant in ("pp256", "w4"):
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    C2 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    hip.gemm(A, B, out, M, N, K, lda=K, ldb=K, ldc=N, mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=rps, C2=C2, ldc2=N,
             variant=hip.GEMM_VARIANT_NAMES[variant])
    torch.cuda.synchronize()
    bad = (C2.float() - raw).abs() > 0.02 * raw.abs().max() + 1e-3
    ref = res.float() + gate.float().repeat_interleave(rps, 0)[:M] * raw.to(torch.bfloat16).float()
    bad2 = ~((out.float() - ref).abs() <= 0.02 * ref.abs().max() + 1e-3)
    print(variant, "raw copy bad:", int(bad.sum()), "out bad:", int(bad2.sum()), "nan in C2:", int(torch.isnan(C2).sum()))
    for name, b in (("raw", bad), ("out", bad2)):
        if b.any():
            idx = b.nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            print(" ", name, "rows min/max", int(rows.min()), int(rows.max()), "cols min/max", int(cols.min()), int(cols.max()))
            print("  distinct row%256:", sorted(set((rows % 256).tolist()))[:40], " distinct col%256 // 8:", sorted(set(((cols % 256) // 8).tolist()))[:40])
            print("  distinct tiles (m/256, n/256):", sorted(set(zip((rows // 256).tolist(), (cols // 256).tolist())))[:20])
