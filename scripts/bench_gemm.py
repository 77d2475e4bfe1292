"""Micro-benchmark of the MFMA GEMM at the MicroDiT-XL/2 shapes (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd import hip

dev = "cuda"
shapes = [  # (M, N, K, akc, bkc, mode, ksplit, label)
    (16384, 3072, 1024, 1, 1, hip.EPI_STORE_BF16, 1, "bb qkv fwd"),
    (16384, 1024, 1024, 1, 1, hip.EPI_STORE_BF16, 1, "bb proj fwd"),
    (65536, 2304, 768, 1, 1, hip.EPI_STORE_BF16, 1, "mixer qkv fwd"),
    (19712, 2048, 1024, 1, 1, hip.EPI_STORE_BF16, 1, "caption kv fwd"),
    (16384, 1024, 3072, 1, 0, hip.EPI_STORE_BF16, 1, "bb qkv dgrad (NN)"),
    (3072, 1024, 16384, 0, 0, hip.EPI_ATOMIC_F32, 8, "bb qkv wgrad (TN, splitk8)"),
    (8192, 8192, 8192, 1, 1, hip.EPI_STORE_BF16, 1, "8k cube"),
    (256, 6144, 1024, 1, 1, hip.EPI_STORE_BF16, 1, "adaLN"),
]
for M, N, K, akc, bkc, mode, ks, label in shapes:
    A = torch.randn((M, K) if akc else (K, M), device=dev).to(torch.bfloat16)
    B = torch.randn((N, K) if bkc else (K, N), device=dev).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if mode == hip.EPI_ATOMIC_F32 else torch.bfloat16)
    def run():
        hip.gemm(A, B, C, M, N, K, lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc,
                 b_kcontig=bkc, mode=mode, ksplit=ks)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{label:32s} M={M:6d} N={N:5d} K={K:6d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
