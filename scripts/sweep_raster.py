"""Sweep md_gemm_args.raster_group_n (column-tiles per L2 raster group of the persistent GEMM) on the heaviest XL/2 shapes:
python scripts/sweep_raster.py [mb]   (TFLOP/s per group width; 0 = the library's rule)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from micro_diffusion_amd import hip             # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
VARIANT = hip.GEMM_VARIANT_NAMES[sys.argv[2]] if len(sys.argv) > 2 else hip.GEMM_PP256      # python scripts/sweep_raster.py 1024 w4
f = mb // 256
SH = [(16384 * f, 1024, 1024, 1, 1), (16384 * f, 1024, 1024, 1, 0), (19712 * f, 2048, 1024, 1, 1), (65536 * f, 768, 768, 1, 1), (65536 * f, 2304, 768, 1, 1),
      (65536 * f, 768, 2304, 1, 0), (16384 * f, 3072, 1024, 1, 1), (16384 * f, 1024, 3072, 1, 0), (65536 * f, 4096, 768, 1, 1), (16384 * f, 5632, 1024, 1, 1),
      (16384 * f, 2688, 1024, 1, 1)]
dev = "cuda"
print(f"# microbatch {mb}; columns: raster_group_n = 0 (library rule), 1, 2, 3, 4, 6, 8, 12, 16")
for M, N, K, akc, bkc in SH:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn((N, K) if bkc else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for g in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        if g > (N + 255) // 256:
            row.append("   -")
            continue
        def run():
            hip.gemm(A, B, C, M, N, K, lda=K, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc, variant=VARIANT, raster_group_n=g)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        row.append(f"{2.0 * M * N * K / best / 1e9:4.0f}")
    print(f"{M:7d} x {N:5d} x {K:5d} {'NT' if bkc else 'NN'}  " + " ".join(row), flush=True)
