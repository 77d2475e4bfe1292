"""TFLOP/s of the w4 GEMM (and pp256 beside it) on a few K-contiguous x K-contiguous shapes, for A/B runs of library builds
(MICRODIT_LIB=scratch_libs/lib_w4_<name>.so python scripts/bench_w4.py).  Random operands, best of 3 x 8 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_diffusion_amd import hip  # noqa: E402

shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (65536, 1024, 1024), (65536, 3072, 1024), (262144, 768, 768), (16384, 1024, 1024)]
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["w4", "pp256"]
bkc = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 0: B K-strided ([K, N] row-major: dgrads, expert weights)
dev = "cuda"
out = []
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn((N, K) if bkc else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for v in variants:
        kw = dict(lda=K, ldb=K if bkc else N, ldc=N, b_kcontig=bkc, variant=hip.GEMM_VARIANT_NAMES[v])
        hip.gemm(A, B, C, M, N, K, **kw)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                hip.gemm(A, B, C, M, N, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8)
        row.append(f"{v} {2.0 * M * N * K / best / 1e9:6.0f}")
    out.append(f"{M}x{N}x{K}: " + "  ".join(row))
print(os.environ.get("MICRODIT_LIB", "library"), "NT" if bkc else "NN", "|", " | ".join(out))
