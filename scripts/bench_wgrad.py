"""Weight-gradient GEMMs (TN layout: both operands K-strided, fp32 split-K slices + reduce) of one XL/2 microbatch of
1024 images: TFLOP/s per (shape, split-K factor) for one GEMM kernel (--variant; default = the library's own choice).
Usage: python scripts/bench_wgrad.py [--variant paced256|pp256|...] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from micro_diffusion_amd import hip             # noqa: E402

# (out_rows, out_cols, contraction, batch, launches per microbatch)
SHAPES = [
    (1024, 1024, 65536, 1, 66), (2048, 1024, 78848, 1, 28), (768, 768, 262144, 1, 18), (2304, 768, 262144, 1, 6),
    (4096, 768, 262144, 1, 3), (768, 2048, 262144, 1, 3), (3072, 1024, 65536, 1, 7), (5376, 1024, 65536, 1, 7),
    (1024, 2688, 65536, 1, 7), (768, 3072, 65536, 8, 3), (3072, 768, 65536, 8, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--mb", type=int, default=1024, help="microbatch the contraction lengths are scaled to (1024 = table as is)")
    ap.add_argument("--variant", default="auto")
    a = ap.parse_args()
    var = hip.GEMM_VARIANT_NAMES[a.variant]
    L = hip.lib()
    ws = torch.empty(64 << 20, device="cuda")       # 256 MiB of fp32 slices
    print(f"# variant {a.variant}")
    for (M, N, K, batch, cnt) in SHAPES:
        K = K * a.mb // 1024
        A = torch.randn(batch, K, M, device="cuda").bfloat16()
        B = torch.randn(batch, K, N, device="cuda").bfloat16()
        out = torch.zeros(batch, M, N, device="cuda")
        t128 = ((M + 127) // 128) * ((N + 127) // 128) * batch
        t256 = ((M + 255) // 256) * ((N + 255) // 256) * batch
        cands = sorted({max(1, 768 // t128), max(1, 256 // t256), max(1, 512 // t256), max(1, 768 // t256)} | {k for k in (2, 4, 8, 16, 32) if 128 <= k * t256 <= 1024})
        row = []
        for ks in cands:
            if ks * M * N * batch > ws.numel():
                continue
            def run():
                if ks == 1:
                    hip.gemm(A, B, out, M, N, K, lda=M, ldb=N, ldc=N, a_kcontig=0, b_kcontig=0, mode=hip.EPI_ACCUM_F32,
                             batch=batch, sA=K * M, sB=K * N, sC=M * N, variant=var)
                else:
                    hip.gemm(A, B, ws, M, N, K, lda=M, ldb=N, ldc=N, a_kcontig=0, b_kcontig=0, mode=hip.EPI_STORE_F32,
                             batch=batch, sA=K * M, sB=K * N, sC=ks * M * N, sSplit=M * N, ksplit=ks, variant=var)
                    hip.check(L.md_splitk_reduce(ws.data_ptr(), out.data_ptr(), M, N, N, M * N, ks, batch, 1, hip.stream_ptr()), "reduce")
            try:
                run()
            except RuntimeError:      # the forced kernel refuses this problem
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            row.append(f"ks={ks:3d}: {ms:7.3f} ms {2.0 * M * N * K * batch / ms / 1e9:6.0f} TF/s")
        print(f"{M:5d} x{N:5d} K={K:6d} b={batch} (x{cnt:2d})  " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
