"""Attention fwd+bwd micro-benchmark at the MicroDiT-XL/2 shapes: python scripts/bench_attn.py [iters] [batch]
bwd: the library's choice, the fused single launch in its single-phase and two-phase forms (the latter for Sq, Skv <= 96) and
the dQ + dK/dV kernel pair; algorithmic HBM bytes / time alongside."""
import os, sys, math
from ctypes import byref
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd import hip
L = hip.lib(); dev = "cuda"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
BB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
shapes = [("mixer self  S=256 H=12", BB, 12, 256, 256, True), ("backbone self S=64 H=16", BB, 16, 64, 64, True),
          ("cross Sq=64 Skv=77 H=16", BB, 16, 64, 77, False), ("caption self S=77 H=16", BB, 16, 77, 77, True),
          ("mixer cross 256x77 H=12", BB, 12, 256, 77, False),
          # res_512_pretrain (64 x 64 latents: 1024 mixer tokens), a 256-image microbatch = BB / 4 samples
          ("res512 mixer self S=1024", max(1, BB // 4), 12, 1024, 1024, True), ("res512 mixer cross 1024x77", max(1, BB // 4), 12, 1024, 77, False)]
for name, B, H, Sq, Skv, packed in shapes:
    hd, hid = 64, H * 64
    if packed:
        qkv = torch.randn(B, Sq, 3 * hid, device=dev).bfloat16(); dqkv = torch.zeros_like(qkv)
        q, k, v, dq, dk, dv = qkv, qkv[..., hid:], qkv[..., 2 * hid:], dqkv, dqkv[..., hid:], dqkv[..., 2 * hid:]
        ld = (3 * hid,) * 3
    else:
        qb = torch.randn(B, Sq, hid, device=dev).bfloat16(); kv = torch.randn(B, Skv, 2 * hid, device=dev).bfloat16()
        dqb, dkv = torch.zeros_like(qb), torch.zeros_like(kv)
        q, k, v, dq, dk, dv = qb, kv, kv[..., hid:], dqb, dkv, dkv[..., hid:]
        ld = (hid, 2 * hid, 2 * hid)
    o = torch.zeros(B, Sq, hid, device=dev, dtype=torch.bfloat16); do = torch.randn_like(o)
    lse = torch.zeros(B, H, Sq, device=dev); delta = torch.zeros(B, H, Sq, device=dev)
    a = hip.AttnArgs(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(),
                     dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Sq, Skv, ld[0], ld[1], ld[2], hid,
                     Sq * ld[0], Skv * ld[1], Skv * ld[2], Sq * hid, ld[0], ld[1], ld[2], hid, Sq * ld[0], Skv * ld[1],
                     Skv * ld[2], Sq * hid, 1 / math.sqrt(hd), hd, 0)
    st = hip.stream_ptr()
    elt = B * H * hd * 2
    for fn, label, mult, split, byt in ((L.md_attn_fwd, "fwd        ", 4, 0, elt * (2 * Sq + 2 * Skv)), (L.md_attn_bwd, "bwd auto   ", 10, 0, elt * (4 * Sq + 4 * Skv)),
                                        (L.md_attn_bwd, "bwd 1-phase", 10, 2, elt * (4 * Sq + 4 * Skv)), (L.md_attn_bwd, "bwd 2-phase", 10, 3, elt * (4 * Sq + 4 * Skv)),
                                        (L.md_attn_bwd, "bwd 2ph+spl", 10, 4, elt * (4 * Sq + 4 * Skv)),
                                        (L.md_attn_bwd, "bwd pair   ", 10, 1, elt * (4 * Sq + 4 * Skv)),
                                        (L.md_attn_bwd, "bwd stream ", 10, 5, elt * (4 * Sq + 4 * Skv))):
        a.bwd_split = split
        if fn(byref(a), st) == -1:
            continue                      # a forced form that does not cover this shape
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn(byref(a), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name:26s} {label}: {us:8.1f} us  {mult*B*H*Sq*Skv*hd/us/1e6:7.1f} TFLOP/s  {byt/us/1e6:6.2f} TB/s algorithmic", flush=True)
