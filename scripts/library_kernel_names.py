"""Which kernels does the library GEMM of this image run on the MicroDiT shapes?  (measurement aid for scripts/bench_gemm_vs_library.py:
run under `rocprofv3 --kernel-trace --stats`; hipBLASLt kernel names spell out macro tile, MFMA shape, wave tile, prefetch depths.)"""
import torch
dev = "cuda"
for (M, N, K, lay) in [(65536, 1024, 1024, "NT"), (65536, 1024, 1024, "NN"), (16384, 1024, 1024, "NT"), (65536, 2304, 768, "NT"), (65536, 768, 2304, "NN"),
                       (1024, 1024, 65536, "TN")]:
    if lay == "TN":
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        f = lambda: torch.matmul(A.t(), B)
    elif lay == "NT":
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
        f = lambda: torch.matmul(A, B.t())
    else:
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        f = lambda: torch.matmul(A, B)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    print(M, N, K, lay, flush=True)
