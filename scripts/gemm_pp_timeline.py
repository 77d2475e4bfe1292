"""Where does a pp256 launch spend its time?  Per-workgroup stamps (md_gemm_args.timeline -> gemm_pp.hip): entry, first k-tile
landed, the end of every tile's k-loop, exit (stores drained), plus wall clocks at entry / exit.
Usage: python scripts/gemm_pp_timeline.py M N K akc bkc [mode: bf16|res|f32|gelu] [batch] [ksplit]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402
from micro_diffusion_amd import hip   # noqa: E402

M, N, K, akc, bkc = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else "bf16"
batch = int(sys.argv[7]) if len(sys.argv) > 7 else 1
ks = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = "cuda"
A = torch.randn((batch, M, K) if akc else (batch, K, M), device=dev).bfloat16()
B = (torch.randn((batch, N, K) if bkc else (batch, K, N), device=dev) * 0.05).bfloat16()
f32 = mode == "f32"
C = torch.zeros(batch * ks, M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
res = torch.randn(M, N, device=dev).bfloat16() if mode == "res" else None
gate = torch.randn(M // 64, N, device=dev).bfloat16() if mode == "res" else None
kw = dict(lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc, variant=hip.GEMM_PP256, batch=batch,
          sA=M * K, sB=N * K, sC=ks * M * N, ksplit=ks)
if mode == "res":
    kw.update(mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=64)
elif f32:
    kw.update(mode=hip.EPI_STORE_F32, sSplit=M * N)
elif mode == "gelu":
    kw.update(act=hip.ACT_GELU_ERF)


def run(timeline=None):
    hip.gemm(A, B, C, M, N, K, timeline=timeline, **kw)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
flops = 2.0 * M * N * K * batch
print(f"shape {M}x{N}x{K} batch {batch} ksplit {ks} layout {akc}{bkc} epilogue {mode}: {us:.1f} us per launch back to back, "
      f"{flops / us / 1e6:.0f} TFLOP/s (un-instrumented)")
tl = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
run(tl)
torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(-1, 16)
t = t[t[:, 0] != 0]
n = len(t)
wall0, wall1 = t[:, 1].astype(np.float64) / 100.0, t[:, 12].astype(np.float64) / 100.0      # us (100 MHz)
clk = t.astype(np.float64)
span_us = wall1.max() - wall0.min()
ghz = ((clk[:, 11] - clk[:, 0]) / np.maximum(wall1 - wall0, 1e-9)).mean() / 1e3
ntile = int(((t[:, 3:11] != 0).sum(1)).max())
print(f"instrumented: {n} workgroups, span first entry -> last exit {span_us:.1f} us ({flops / span_us / 1e6:.0f} TFLOP/s inside the kernel); "
      f"launch-to-launch minus span = {us - span_us:.1f} us; shader clock {ghz:.2f} GHz; up to {ntile} tiles / workgroup")


def q(x):
    return f"mean {x.mean():7.2f}  min {x.min():7.2f}  p50 {np.percentile(x, 50):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f}"


cyc2us = 1.0 / (ghz * 1e3)
print(f"entry skew      (us after the first entry): {q(wall0 - wall0.min())}")
print(f"exit skew       (us before the last exit) : {q(wall1.max() - wall1)}")
print(f"prologue        (entry -> first k-tile)   : {q((clk[:, 2] - clk[:, 0]) * cyc2us)}")
prev = clk[:, 2]
ideal = 2.0 * 256 * 256 * (K // ks) / (4 * 32768 / 32) * cyc2us
for i in range(ntile):
    have = t[:, 3 + i] != 0
    d = (clk[have, 3 + i] - prev[have]) * cyc2us
    print(f"tile {i} k-loop (+ previous tile's epilogue)   : {q(d)}   [{have.sum()} workgroups; MFMA-bound {ideal:.2f} us at this clock]")
    prev = np.where(have, clk[:, 3 + i], prev)
print(f"last epilogue + store drain               : {q((clk[:, 11] - prev) * cyc2us)}")
