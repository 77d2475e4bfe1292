"""Experiment (round 4): does an HBM-bound kernel (SwiGLU backward, 1.85 GB of traffic, no MFMA) overlap with an MFMA-bound one (weight-gradient
GEMMs, persistent, md_gemm_args.cu_limit leaving CUs free) when the two run on separate HIP streams?  The backward of a DiT block alternates the two
kinds on its critical path while its weight gradients are off it; if the pair finishes well before the sum of its parts, a second stream for the
weight gradients is worth building.

    python scripts/exp_overlap_hbm_mfma.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from micro_diffusion_amd import hip  # noqa: E402

dev = "cuda"
L = hip.lib()
M, f = 65536, 2816
N = K = 1024
torch.manual_seed(0)
h12 = torch.randn(M, 2 * f, device=dev).bfloat16()
da = torch.randn(M, f, device=dev).bfloat16()
dh = torch.empty_like(h12)
dy = torch.randn(M, N, device=dev).bfloat16()
x = torch.randn(M, K, device=dev).bfloat16()
ks = 16
ws = torch.empty(ks, N, K, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NG = 3          # weight-gradient launches per SwiGLU backward: ~equal durations


def hbm(stream):
    hip.check(L.md_swiglu_bwd(da.data_ptr(), f, h12.data_ptr(), 2 * f, dh.data_ptr(), 2 * f, M, f, stream.cuda_stream), "swiglu_bwd")


def mfma(stream, lim):
    for _ in range(NG):
        hip.gemm(dy, x, ws, N, K, M, lda=N, ldb=K, ldc=K, a_kcontig=False, b_kcontig=False, mode=hip.EPI_STORE_F32, ksplit=ks, sSplit=N * K, sC=ks * N * K,
                 variant=hip.GEMM_PP256, stream=stream.cuda_stream, cu_limit=lim)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / reps)
    return best


cur = torch.cuda.current_stream()
t_h = timed(lambda: hbm(cur))
t_m = timed(lambda: mfma(cur, 0))
t_seq = timed(lambda: (hbm(cur), mfma(cur, 0)))
print(f"SwiGLU backward alone {t_h:.1f} us | {NG} weight-gradient GEMMs alone {t_m:.1f} us | one after the other {t_seq:.1f} us")


def both(lim, hbm_first):
    def run():
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        if hbm_first:
            hbm(s1)
            mfma(s2, lim)
        else:
            mfma(s2, lim)
            hbm(s1)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    return run


for lim in (0, 240, 224, 192, 160, 128):
    for first in (True, False):
        t = timed(both(lim, first))
        print(f"  two streams, GEMM cu_limit {lim or 256:3d}, {'HBM kernel' if first else 'GEMMs'} enqueued first: {t:.1f} us  (x{t_seq / t:.3f} vs sequential)")
