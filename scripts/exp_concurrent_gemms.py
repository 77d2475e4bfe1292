"""Experiment (round 4): do two INDEPENDENT persistent GEMMs finish sooner side by side on half the chip each (two HIP streams,
md_gemm_args.cu_limit = 128) than one after the other on all 256 CUs?  The backward of a DiT block has such pairs -- the dgrad
chain and the weight gradients of the same layer -- and at the 256-image per-rank shape of an 8-GPU run every N = 1024 activation
GEMM has exactly one tile per workgroup (no epilogue overlap, 0.31-0.35 of peak); on 128 CUs the same launch has two.

    python scripts/exp_concurrent_gemms.py [tokens]      (tokens = 16384: microbatch 256; 65536: microbatch 1024)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from micro_diffusion_amd import hip  # noqa: E402

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = K = 1024
torch.manual_seed(0)
dy = torch.randn(M, N, device=dev).bfloat16()
x = torch.randn(M, K, device=dev).bfloat16()
w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
ks = 16
ws = torch.empty(ks, N, K, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def dgrad(stream, lim):      # dx = dy @ W (NN, bf16 out): M/256 x 4 tiles
    hip.gemm(dy, w, dx, M, K, N, lda=N, ldb=K, ldc=K, a_kcontig=True, b_kcontig=False, variant=hip.GEMM_PP256, stream=stream.cuda_stream, cu_limit=lim)


def wgrad(stream, lim):      # dW slices = dy^T x (TN, fp32 slices): 16 tiles x ks 16 = 256 items
    hip.gemm(dy, x, ws, N, K, M, lda=N, ldb=K, ldc=K, a_kcontig=False, b_kcontig=False, mode=hip.EPI_STORE_F32, ksplit=ks, sSplit=N * K, sC=ks * N * K,
             variant=hip.GEMM_PP256, stream=stream.cuda_stream, cu_limit=lim)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            fn()
        t1.record()
        torch.cuda.synchronize()
        best = min(best, t0.elapsed_time(t1) * 1e3 / reps)
    return best


def sequential():
    cur = torch.cuda.current_stream()
    dgrad(cur, 0)
    wgrad(cur, 0)


def concurrent(la, lb):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        dgrad(s1, la)
        wgrad(s2, lb)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    return run


fl = 2.0 * M * N * K * 2
t_d = timed(lambda: dgrad(torch.cuda.current_stream(), 0))
t_w = timed(lambda: wgrad(torch.cuda.current_stream(), 0))
t_seq = timed(sequential)
print(f"tokens {M}: dgrad alone {t_d:.1f} us | wgrad(ks {ks}) alone {t_w:.1f} us | one after the other {t_seq:.1f} us ({fl / t_seq / 1e6:.0f} TFLOP/s)")
for la, lb in ((128, 128), (160, 96), (96, 160), (0, 0)):
    t = timed(concurrent(la, lb))
    print(f"  side by side on two streams, cu_limit {la or 256} / {lb or 256}: {t:.1f} us ({fl / t / 1e6:.0f} TFLOP/s, x{t_seq / t:.3f})")
