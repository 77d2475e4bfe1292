"""What the q / k / v LAYOUT costs the attention kernels: the same kernels on the same (batch, head) problems, once with the packed
row-major layout the engine uses (a head is a 128-byte slice of a 2-6 KB row: 64..256 separate 128-byte pieces per tile) and once
head-major (every (batch, head) tile one contiguous block), which md_attn_args can already express: B' = B * H "batches" of ONE
head with row pitch head_dim.  Nothing else differs, so the ratio is the price of the strided access.
    python scripts/bench_attn_layout.py [iters] [batch]"""
import math
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from micro_diffusion_amd import hip  # noqa: E402

L = hip.lib()
dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
BB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
shapes = [("mixer self  S=256 H=12", BB, 12, 256, 256, True), ("backbone self S=64 H=16", BB, 16, 64, 64, True),
          ("cross Sq=64 Skv=77 H=16", BB, 16, 64, 77, False), ("mixer cross 256x77 H=12", BB, 12, 256, 77, False)]
hd = 64
st = hip.stream_ptr()
for name, B, H, Sq, Skv, packed in shapes:
    hid = H * hd
    res = {}
    for layout in ("row-major (engine)", "head-major"):
        if layout.startswith("row"):
            if packed:
                qkv = torch.randn(B, Sq, 3 * hid, device=dev).bfloat16()
                dqkv = torch.zeros_like(qkv)
                q, k, v, dq, dk, dv = qkv, qkv[..., hid:], qkv[..., 2 * hid:], dqkv, dqkv[..., hid:], dqkv[..., 2 * hid:]
                ld = (3 * hid,) * 3
            else:
                qb = torch.randn(B, Sq, hid, device=dev).bfloat16()
                kv = torch.randn(B, Skv, 2 * hid, device=dev).bfloat16()
                dqb, dkv = torch.zeros_like(qb), torch.zeros_like(kv)
                q, k, v, dq, dk, dv = qb, kv, kv[..., hid:], dqb, dkv, dkv[..., hid:]
                ld = (hid, 2 * hid, 2 * hid)
            o = torch.zeros(B, Sq, hid, device=dev, dtype=torch.bfloat16)
            do = torch.randn_like(o)
            Bx, Hx, ldo = B, H, hid
            sq, sk, sv, so = Sq * ld[0], Skv * ld[1], Skv * ld[2], Sq * hid
        else:
            Bx, Hx, ldo = B * H, 1, hd
            q = torch.randn(Bx, Sq, hd, device=dev).bfloat16()
            k = torch.randn(Bx, Skv, hd, device=dev).bfloat16()
            v = torch.randn(Bx, Skv, hd, device=dev).bfloat16()
            dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
            o = torch.zeros(Bx, Sq, hd, device=dev, dtype=torch.bfloat16)
            do = torch.randn_like(o)
            ld = (hd, hd, hd)
            sq, sk, sv, so = Sq * hd, Skv * hd, Skv * hd, Sq * hd
        lse = torch.zeros(Bx, Hx, Sq, device=dev)
        delta = torch.zeros(Bx, Hx, Sq, device=dev)
        a = hip.AttnArgs(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                         dv.data_ptr(), delta.data_ptr(), Bx, Hx, Sq, Skv, ld[0], ld[1], ld[2], ldo, sq, sk, sv, so, ld[0], ld[1], ld[2], ldo,
                         sq, sk, sv, so, 1 / math.sqrt(hd), hd, 0)
        elt = B * H * hd * 2
        for fn, label, byt in ((L.md_attn_fwd, "fwd", elt * (2 * Sq + 2 * Skv)), (L.md_attn_bwd, "bwd", elt * (4 * Sq + 4 * Skv))):
            hip.check(fn(byref(a), st), label)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(byref(a), st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            res[(layout, label)] = (us, byt / us / 1e6)
        del q, k, v, dq, dk, dv, o, do
        torch.cuda.empty_cache()
    for label in ("fwd", "bwd"):
        (u0, t0), (u1, t1) = res[("row-major (engine)", label)], res[("head-major", label)]
        print(f"{name:26s} {label}: row-major {u0:8.1f} us {t0:5.2f} TB/s | head-major {u1:8.1f} us {t1:5.2f} TB/s | x{u0 / u1:4.2f}", flush=True)
