"""Attention launches for rocprofv3 --pmc passes: python scripts/pmc_attn.py [B] [small]  (S = 1024 / 1024 x 77 / 256 self-attention, forward +
the library's backward, 3 launches each; kernels are told apart by name in the counter CSV).  "small": the backbone's 64-row shapes
instead (64 x 64 self-attention and 64 x 77 cross-attention, 16 heads, 4 B samples)."""
import math
import os
import sys
from ctypes import byref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                    # noqa: E402
from micro_diffusion_amd import hip             # noqa: E402

L, dev = hip.lib(), "cuda"
BB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = ((BB, 12, 1024, 1024, True), (BB, 12, 1024, 77, False), (BB * 4, 12, 256, 256, True))
if len(sys.argv) > 2 and sys.argv[2] == "small":
    SHAPES = ((BB * 4, 16, 64, 64, True), (BB * 4, 16, 64, 77, False))
for B, H, Sq, Skv, packed in SHAPES:
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
    a = hip.AttnArgs(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                     delta.data_ptr(), B, H, Sq, Skv, ld[0], ld[1], ld[2], hid, Sq * ld[0], Skv * ld[1], Skv * ld[2], Sq * hid, ld[0], ld[1], ld[2], hid,
                     Sq * ld[0], Skv * ld[1], Skv * ld[2], Sq * hid, 1 / math.sqrt(hd), hd, 0)
    st = hip.stream_ptr()
    for _ in range(3):
        hip.check(L.md_attn_fwd(byref(a), st), "fwd")
        hip.check(L.md_attn_bwd(byref(a), st), "bwd")
    torch.cuda.synchronize()
print("done")
