"""Where does a GEMM launch spend its time?  Per-workgroup shader-clock stamps (md_gemm_args.timeline) of one shape:
prologue (entry -> first tile landed), k-loop, epilogue (incl. store drain), and the gap between consecutive workgroups
on the same CU slot.  Usage: python scripts/gemm_timeline.py M N K akc bkc [mode: bf16|res|f32] [variant: auto|reg128|dma128|paced256]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402
from micro_diffusion_amd import hip   # noqa: E402

M, N, K, akc, bkc = [int(v) for v in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else "bf16"
variant = sys.argv[7] if len(sys.argv) > 7 else "auto"
dev = "cuda"
A = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
B = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
C = torch.zeros(M, N, device=dev, dtype=torch.float32 if mode == "f32" else torch.bfloat16)
res = torch.randn(M, N, device=dev).bfloat16() if mode == "res" else None
gate = torch.randn(M // 64, N, device=dev).bfloat16() if mode == "res" else None
kw = dict(lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc, variant=hip.GEMM_VARIANT_NAMES[variant])
if mode == "res":
    kw.update(mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=64)
elif mode == "f32":
    kw.update(mode=hip.EPI_STORE_F32)


def run(timeline=None):
    hip.gemm(A, B, C, M, N, K, timeline=timeline, **kw)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"shape {M}x{N}x{K} layout {akc}{bkc} epilogue {mode} variant {variant}: "
      f"{ms * 1e3:.1f} us, {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s (un-instrumented)")

nblk_max = ((M + 127) // 128) * ((N + 127) // 128)
tl = torch.zeros(nblk_max * 8, dtype=torch.int64, device=dev)
e0.record()
run(tl)
e1.record()
torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
n = len(t)
clk = t[:, :4].astype(np.float64)
span_cyc = clk[:, 3].max() - clk[:, 0].min()
wall = t[:, 4].astype(np.float64)
span_wall_us = (wall.max() - wall.min()) / 100.0
ghz = span_cyc / max(span_wall_us, 1e-9) / 1e3
print(f"instrumented launch: {e0.elapsed_time(e1) * 1e3:.1f} us, {n} workgroups, span {span_cyc:.0f} cycles = {span_wall_us:.1f} us "
      f"-> shader clock {ghz:.2f} GHz")
pro, loop, epi = clk[:, 1] - clk[:, 0], clk[:, 2] - clk[:, 1], clk[:, 3] - clk[:, 2]
tot = clk[:, 3] - clk[:, 0]


def q(x):
    return f"mean {x.mean():8.0f}  p10 {np.percentile(x, 10):8.0f}  p50 {np.percentile(x, 50):8.0f}  p90 {np.percentile(x, 90):8.0f}"


print(f"prologue  (cycles): {q(pro)}   {100 * pro.sum() / tot.sum():5.1f} % of workgroup time")
print(f"k-loop    (cycles): {q(loop)}   {100 * loop.sum() / tot.sum():5.1f} %")
print(f"epilogue  (cycles): {q(epi)}   {100 * epi.sum() / tot.sum():5.1f} %")
hw = t[:, 5]
hwid = hw & 0xFFFFFFFF
xcc = hw >> 32
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
cu = ((xcc & 0xF) << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xF)
cus = np.unique(cu)
print(f"distinct CUs seen: {len(cus)}; workgroups per CU: min {min((cu == c).sum() for c in cus)} max {max((cu == c).sum() for c in cus)}")
# concurrency and gaps per CU
gaps, conc = [], []
for c in cus:
    rows = clk[cu == c]
    order = np.argsort(rows[:, 0])
    rows = rows[order]
    # residency: average number of co-resident workgroups over the CU's busy span
    busy = rows[:, 3].max() - rows[:, 0].min()
    conc.append((rows[:, 3] - rows[:, 0]).sum() / busy)
    ends = np.sort(rows[:, 3])
    starts = rows[:, 0]
    k = int(round(conc[-1]))
    if len(rows) > k > 0:
        g = starts[k:] - ends[:len(rows) - k]
        gaps.extend(g.tolist())
gaps = np.array(gaps) if gaps else np.zeros(1)
print(f"co-resident workgroups per CU: mean {np.mean(conc):.2f}; slot turnaround (end of a workgroup -> entry of the next "
      f"on that CU): {q(gaps)}")
print(f"ideal MFMA time per workgroup at 32 cycles / MFMA / SIMD: {2.0 * M * N * K / n / (4 * 32768 / 32):.0f} cycles of one CU")
