"""Yardstick, measurement only: the library GEMM of this image (hipBLASLt / rocBLAS behind torch.matmul, bf16 in,
fp32 accumulate) against md_gemm_bf16 on the GEMM shapes of the MicroDiT-XL/2 step, plain stores, at the per-microbatch row
counts of microbatch 256 and 1024.  NOT on the product path (the engine never imports torch.matmul); the table tells where the
hand-written kernels have head-room that a library already demonstrates on this chip, and where they do not.

    python scripts/bench_gemm_vs_library.py [--iters 10] > profiles/r5_gemm_vs_hipblaslt.txt

Layouts (md_gemm_args): NT = activations x torch weights [N, K] (forward); NN = dgrad (B stored [K, N]); TN = weight gradient
(both operands stored token-major, contraction over tokens; ours = fp32 split-K slices + md_splitk_reduce as the engine launches
them, the library's = one bf16-output matmul, which is LESS work: it neither accumulates into an fp32 gradient nor keeps fp32)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from micro_diffusion_amd import hip             # noqa: E402

# (rows per 256-image microbatch, N, K, layout, launches per microbatch): profiles/r3_gemm_shapes_mb256.txt, single-problem launches
SHAPES = [
    (16384, 1024, 1024, "NT", 65), (16384, 1024, 1024, "NN", 65), (19712, 2048, 1024, "NT", 28), (65536, 768, 768, "NT", 18),
    (65536, 768, 768, "NN", 18), (65536, 2304, 768, "NT", 6), (65536, 768, 2304, "NN", 6), (65536, 4096, 768, "NT", 3),
    (65536, 768, 4096, "NN", 3), (16384, 3072, 1024, "NT", 8), (16384, 1024, 3072, "NN", 8), (16384, 2688, 1024, "NT", 7),
    (16384, 2304, 1024, "NT", 7), (16384, 1024, 2688, "NN", 7), (16384, 1024, 768, "NT", 10), (16384, 1920, 1024, "NT", 6),
    (16384, 1024, 896, "NT", 7), (16384, 5632, 1024, "NT", 2),
    # weight gradients: (out rows, out cols, tokens per 256-image microbatch)
    (1024, 1024, 16384, "TN", 65), (2048, 1024, 19712, "TN", 28), (768, 768, 65536, "TN", 18), (3072, 1024, 16384, "TN", 8),
    (2304, 768, 65536, "TN", 6),
]


def time_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def ksplit_like_engine(out_rows, out_cols, contraction):
    """DiTEngine._ksplit's rule for pp256-eligible weight gradients (contraction % 128 == 0)."""
    t256 = ((out_rows + 255) // 256) * ((out_cols + 255) // 256)
    units = contraction // 128
    out_us = out_rows * out_cols * 4 / 4e6
    best, best_cost = 1, None
    for ks in range(1, min(64, units) + 1):
        if units % ks or t256 * ks < 192:
            continue
        per_wg = -(-t256 * ks // 256)
        cost = per_wg * (contraction / ks / 64 * 1.9 + 5.0) + (ks + 2) * out_us
        if best_cost is None or cost < best_cost:
            best, best_cost = ks, cost
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    L = hip.lib()
    dev = "cuda"
    ws = torch.empty(96 << 20, device=dev)
    print("# torch.matmul (hipBLASLt / rocBLAS, bf16 -> bf16) vs md_gemm_bf16 (AUTO kernel choice, plain store), best of 3 x %d launches, random operands" % a.iters)
    print("# torch %s, %s" % (torch.__version__, torch.cuda.get_device_name(0)))
    print("# ratio = library TF/s / ours (> 1: the library is faster)")
    print(f"{'mb':>5} {'M':>7} {'N':>6} {'K':>7} {'lay':>3} {'cnt':>4} {'ours us':>9} {'ours TF/s':>9} {'lib us':>9} {'lib TF/s':>9} {'ratio':>6}")
    tot = {256: [0.0, 0.0], 1024: [0.0, 0.0]}
    for mb in (256, 1024):
        f = mb // 256
        for (r, N, K, lay, cnt) in SHAPES:
            if lay == "TN":
                M, Nn, Kk = r, N, K * f
                A = torch.randn(Kk, M, device=dev).to(torch.bfloat16)
                B = torch.randn(Kk, Nn, device=dev).to(torch.bfloat16)
                out = torch.zeros(M, Nn, device=dev)
                ks = ksplit_like_engine(M, Nn, Kk)

                def ours():
                    hip.gemm(A, B, ws, M, Nn, Kk, lda=M, ldb=Nn, ldc=Nn, a_kcontig=0, b_kcontig=0, mode=hip.EPI_STORE_F32, sC=ks * M * Nn,
                             sSplit=M * Nn, ksplit=ks)
                    hip.check(L.md_splitk_reduce(ws.data_ptr(), out.data_ptr(), M, Nn, Nn, M * Nn, ks, 1, 1, hip.stream_ptr()), "reduce")

                def lib():
                    torch.matmul(A.t(), B)
            else:
                M, Nn, Kk = r * f, N, K
                A = torch.randn(M, Kk, device=dev).to(torch.bfloat16)
                C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
                Cl = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
                if lay == "NT":
                    B = (torch.randn(Nn, Kk, device=dev) * 0.05).to(torch.bfloat16)

                    def ours():
                        hip.gemm(A, B, C, M, Nn, Kk, lda=Kk, ldb=Kk, ldc=Nn)

                    def lib():
                        torch.matmul(A, B.t(), out=Cl)
                else:
                    B = (torch.randn(Kk, Nn, device=dev) * 0.05).to(torch.bfloat16)

                    def ours():
                        hip.gemm(A, B, C, M, Nn, Kk, lda=Kk, ldb=Nn, ldc=Nn, b_kcontig=0)

                    def lib():
                        torch.matmul(A, B, out=Cl)
            t_o, t_l = time_us(ours, a.iters), time_us(lib, a.iters)
            fl = 2.0 * M * Nn * Kk
            tot[mb][0] += t_o * cnt
            tot[mb][1] += t_l * cnt
            print(f"{mb:>5} {M:>7} {Nn:>6} {Kk:>7} {lay:>3} {cnt:>4} {t_o:>9.1f} {fl / t_o / 1e6:>9.1f} {t_l:>9.1f} {fl / t_l / 1e6:>9.1f} {t_o / t_l:>6.2f}", flush=True)
            del A, B
    for mb in (256, 1024):
        print(f"# microbatch {mb}: launches-weighted time of the listed shapes -- ours {tot[mb][0] / 1e3:.2f} ms, library {tot[mb][1] / 1e3:.2f} ms "
              f"(library / ours = {tot[mb][1] / tot[mb][0]:.3f})")


if __name__ == "__main__":
    main()
