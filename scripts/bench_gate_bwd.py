"""md_gate_bwd (gated-residual backward: dbr = dx * gate, dgate += sum_t dx * br) over rows_per_block, on the two MicroDiT activation shapes.
Usage: python scripts/bench_gate_bwd.py [--mb 1024]   (the engine's rule: DiTEngine._rows_per_block)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from micro_diffusion_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
L, st, dev = hip.lib(), hip.stream_ptr(), "cuda"
for name, rows, C, rps in (("backbone", args.mb * 64, 1024, 64), ("mixer", args.mb * 256, 768, 256)):
    B = rows // rps
    dx = torch.randn(rows, C, device=dev).bfloat16()
    br = torch.randn(rows, C, device=dev).bfloat16()
    dbr = torch.empty_like(dx)
    mod = (torch.randn(B, 6 * C, device=dev) * 0.3).bfloat16()
    dmod = torch.zeros(B, 6 * C, device=dev)
    r = 4
    while r <= rps:
        fn = lambda: hip.check(L.md_gate_bwd(dx.data_ptr(), br.data_ptr(), mod.data_ptr(), 6 * C, dbr.data_ptr(), dmod.data_ptr(), 6 * C, rows, C, rps, r, st), "gate_bwd")
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print(f"{name} gate_bwd [{rows} x {C}] rows_per_block {r:4d} ({rows // r:6d} workgroups) {us:9.1f} us   {rows * C * 6 / us / 1e6:6.2f} TB/s")
        r *= 2
