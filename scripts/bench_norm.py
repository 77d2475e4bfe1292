"""HBM-rate microbench of the LayerNorm family on the two MicroDiT activation shapes (backbone [mb*64, 1024], patch mixer
[mb*256, 768]): md_ln_fwd (modulated), md_ln_bwd (modulated, dscale / dshift sums), md_qkln_fwd / md_qkln_bwd on the packed
qkv buffer.  Algorithmic bytes / HIP-event time.  Usage: python scripts/bench_norm.py [--mb 1024]   (MICRODIT_LIB selects a build)"""
import argparse
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from micro_diffusion_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rpb-sweep", action="store_true", help="md_ln_bwd with rows_per_block 8 .. rows per sample (the engine's rule: DiTEngine._rows_per_block)")
args = ap.parse_args()
L, st, dev = hip.lib(), hip.stream_ptr(), "cuda"


def timed(fn, nbytes, label):
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
    print(f"{label:44s} {us:9.1f} us   {nbytes / us / 1e6:6.2f} TB/s")


for name, rows, C, rps in (("backbone", args.mb * 64, 1024, 64), ("mixer", args.mb * 256, 768, 256)):
    B = rows // rps
    x = (torch.randn(rows, C, device=dev) * 1.5).to(torch.bfloat16)
    dz = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    out, dx = torch.empty_like(x), torch.empty_like(x)
    w = torch.ones(C, device=dev)
    mod = (torch.randn(B, 6 * C, device=dev) * 0.3).to(torch.bfloat16)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    dmod, dw = torch.zeros(B, 6 * C, device=dev), torch.zeros(C, device=dev)
    a = hip.LnArgs(x.data_ptr(), w.data_ptr(), mod.data_ptr(), mod[:, C:].data_ptr(), None, out.data_ptr(), mean.data_ptr(),
                   rstd.data_ptr(), rows, C, C, C, 6 * C, rps, 0, 1e-6, 0)
    rpb = 64
    while rpb > 4 and (rows + rpb - 1) // rpb < 1024:     # DiTEngine._rows_per_block(rows, rps, 1024)
        rpb //= 2
    b = hip.LnBwdArgs(dz.data_ptr(), dx.data_ptr(), dmod[:, C:].data_ptr(), dmod.data_ptr(), dw.data_ptr(), C, C, 6 * C,
                      min(rpb, rps), 0, 1)
    timed(lambda: hip.check(L.md_ln_fwd(byref(a), st), "ln"), rows * C * 4, f"{name} ln_fwd   [{rows} x {C}]")
    timed(lambda: hip.check(L.md_ln_bwd(byref(a), byref(b), st), "lnb"), rows * C * 6, f"{name} ln_bwd   [{rows} x {C}]")
    if args.rpb_sweep:
        r = 8
        while r <= rps:
            b2 = hip.LnBwdArgs(dz.data_ptr(), dx.data_ptr(), dmod[:, C:].data_ptr(), dmod.data_ptr(), dw.data_ptr(), C, C, 6 * C, r, 0, 1)
            timed(lambda: hip.check(L.md_ln_bwd(byref(a), byref(b2), st), "lnb"), rows * C * 6, f"{name} ln_bwd   rows_per_block {r} ({rows // r} workgroups)")
            r *= 2
    qkv = (torch.randn(rows, 3 * C, device=dev)).to(torch.bfloat16)
    dqkv = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16)
    rq = torch.empty(rows, device=dev)
    timed(lambda: hip.check(L.md_qkln_fwd(qkv.data_ptr(), rows, 3 * C, 0, C, 1, 0, rq.data_ptr(), 1e-6, st), "q"), rows * C * 4,
          f"{name} qkln_fwd [{rows} x {C} of {3 * C}]")
    timed(lambda: hip.check(L.md_qkln_bwd(dqkv.data_ptr(), 3 * C, 0, qkv.data_ptr(), 3 * C, 0, rows, C, 1, 0, 0, rq.data_ptr(), st), "qb"),
          rows * C * 6, f"{name} qkln_bwd [{rows} x {C} of {3 * C}]")
