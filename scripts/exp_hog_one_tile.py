"""The rank-of-8 shape (16,384 x 1024 x 1024: one tile per CU) while 8 CUs are REALLY held (tests/probes: mdp_cu_hog, the stand-in for an
8-channel RCCL kernel): full grid of 256 workgroups vs cu_limit 248 with whole tiles vs cu_limit 248 with the split-K tail.  Median of 9."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from micro_diffusion_amd import hip             # noqa: E402
from tests import probes                        # noqa: E402

dev = "cuda"
for (M, N, K) in ((16384, 1024, 1024), (16384, 3072, 1024), (65536, 768, 768)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tws = torch.empty(256 * 256 * 256, device=dev)
    kw = dict(A=A, B=B, C=C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, variant=hip.GEMM_PP256, tail_ws=tws)
    scratch = torch.zeros(1, device=dev, dtype=torch.int32)
    side = torch.cuda.Stream()

    def timed(hog, **extra):
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            if hog:
                with torch.cuda.stream(side):
                    hip.check(probes.lib().mdp_cu_hog(8, 2000, scratch.data_ptr(), side.cuda_stream), "hog")
                torch.cuda._sleep(200000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.gemm(**kw, **extra)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]
    for _ in range(3):
        hip.gemm(**kw)
    free = timed(False, tail_mode=1)
    print(f"{M} x {N} x {K}: free chip {free:.1f} us | 8 CUs held: full grid {timed(True, tail_mode=1):.1f} | cu_limit 248 whole tiles "
          f"{timed(True, tail_mode=1, cu_limit=248):.1f} | cu_limit 248 + tail {timed(True, tail_mode=0, cu_limit=248):.1f}", flush=True)
