"""Kernel-level parity on the GPU: every HIP kernel family against a plain PyTorch fp32 reference of the same op
(same bf16-rounded inputs).  Tolerances are bf16 output rounding (2^-8 relative) unless noted; index outputs are
compared bit-exact."""
from ctypes import byref

import math
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(t):
    return t.to(torch.bfloat16)


def close(a, b, rel=2e-2, abs_=1e-3, what=""):
    a, b = a.detach().float(), b.detach().float()
    err = (a - b).abs().max().item()
    lim = rel * b.abs().max().item() + abs_
    assert err <= lim, f"{what}: max err {err:.4g} > {lim:.4g} (ref max {b.abs().max().item():.4g})"


def rel_rms(a, b):
    a, b = a.float(), b.float()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("C,rows,rps", [(768, 512, 64), (1024, 154, 77), (256, 64, 16), (128, 96, 32), (2048, 32, 8)])
@pytest.mark.parametrize("mod,act,pos", [(True, 0, False), (False, 0, False), (False, 1, False), (False, 0, True)])
def test_ln_fwd_bwd(hip, C, rows, rps, mod, act, pos):
    torch.manual_seed(C + rows + mod)
    L = hip.lib()
    st = hip.stream_ptr()
    B = rows // rps
    x = bf(torch.randn(rows, C, device=DEV) * 1.5 + 0.3)
    w = (1 + 0.2 * torch.randn(C, device=DEV)).float()
    shift = bf(torch.randn(B, 3 * C, device=DEV) * 0.3)
    scale = bf(torch.randn(B, 3 * C, device=DEV) * 0.3)
    posv = torch.randn(rps, C, device=DEV) if pos else None
    out = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    a = hip.LnArgs(x.data_ptr(), w.data_ptr(), shift[:, C:].data_ptr() if mod else None,
                   scale[:, 2 * C:].data_ptr() if mod else None, posv.data_ptr() if pos else None, out.data_ptr(),
                   mean.data_ptr(), rstd.data_ptr(), rows, C, C, C, 3 * C, rps, rps if pos else 0, 1e-6, act)
    hip.check(L.md_ln_fwd(byref(a), st), "ln_fwd")
    # reference
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    shr = shift[:, C:2 * C].float().requires_grad_(True)
    scr = scale[:, 2 * C:].float().requires_grad_(True)
    xin = xr
    if pos:
        xin = xin + posv.repeat(B, 1)
    if act:
        xin = F.gelu(xin, approximate="tanh")
    y = F.layer_norm(xin, (C,), wr, None, 1e-6)
    if mod:
        y = y * (1 + scr.repeat_interleave(rps, 0)) + shr.repeat_interleave(rps, 0)
    torch.cuda.synchronize()
    close(out, y, what="ln fwd")
    # backward
    dz = bf(torch.randn(rows, C, device=DEV))
    y.backward(dz.float())
    dx = bf(torch.randn(rows, C, device=DEV))
    dx0 = dx.clone()
    dmod = torch.zeros(B, 6 * C, device=DEV)
    dw = torch.zeros(C, device=DEV)
    scratch = torch.zeros(B, C, device=DEV)      # plain LN: zeroed per-sample scratch for the weight-grad finish
    b = hip.LnBwdArgs(dz.data_ptr(), dx.data_ptr(), dmod[:, C:].data_ptr() if mod else scratch.data_ptr(),
                      dmod[:, 0:].data_ptr() if mod else None, dw.data_ptr(), C, C, 6 * C if mod else C, 16, 1,
                      1 if mod else 0)
    hip.check(L.md_ln_bwd(byref(a), byref(b), st), "ln_bwd")
    torch.cuda.synchronize()
    close(dx.float() - dx0.float(), xr.grad, rel=3e-2, what="ln dx")
    close(dw, wr.grad, rel=2e-2, what="ln dw")
    if mod:
        close(dmod[:, C:2 * C], scr.grad, rel=2e-2, what="dscale")
        close(dmod[:, :C], shr.grad, rel=2e-2, what="dshift")


@pytest.mark.parametrize("width", [512, 768, 1024, 128])
def test_qkln(hip, width):
    torch.manual_seed(width)
    L, st = hip.lib(), hip.stream_ptr()
    rows, ld = 200, 3 * width
    buf = bf(torch.randn(rows, ld, device=DEV) * 2 + 0.5)
    ref_in = buf.float().clone().requires_grad_(True)
    rstd = torch.empty(2, rows, device=DEV)
    work = buf.clone()
    hip.check(L.md_qkln_fwd(work.data_ptr(), rows, ld, 0, width, 1, 0, rstd[0].data_ptr(), 1e-6, st), "qkln q")
    hip.check(L.md_qkln_fwd(work.data_ptr(), rows, ld, width, width, 1, 0, rstd[1].data_ptr(), 1e-6, st), "qkln k")
    work2, rstd2 = buf.clone(), torch.empty(2, rows, device=DEV)          # both halves in ONE launch: bit-identical to the two
    hip.check(L.md_qkln_fwd(work2.data_ptr(), rows, ld, 0, width, 2, width, rstd2.data_ptr(), 1e-6, st), "qkln q+k")
    torch.cuda.synchronize()
    assert torch.equal(work2, work) and torch.equal(rstd2, rstd)
    q = F.layer_norm(ref_in[:, :width], (width,), None, None, 1e-6)
    k = F.layer_norm(ref_in[:, width:2 * width], (width,), None, None, 1e-6)
    torch.cuda.synchronize()
    close(work[:, :width], q, what="qkln q")
    close(work[:, width:2 * width], k, what="qkln k")
    assert torch.equal(work[:, 2 * width:], buf[:, 2 * width:])
    d = bf(torch.randn(rows, ld, device=DEV))
    (q * d[:, :width].float()).sum().backward(retain_graph=True)
    (k * d[:, width:2 * width].float()).sum().backward()
    dwork = d.clone()
    hip.check(L.md_qkln_bwd(dwork.data_ptr(), ld, 0, work.data_ptr(), ld, 0, rows, width, 1, 0, 0, rstd[0].data_ptr(), st), "b")
    hip.check(L.md_qkln_bwd(dwork.data_ptr(), ld, width, work.data_ptr(), ld, width, rows, width, 1, 0, 0, rstd[1].data_ptr(), st), "b")
    dwork2 = d.clone()
    hip.check(L.md_qkln_bwd(dwork2.data_ptr(), ld, 0, work.data_ptr(), ld, 0, rows, width, 2, width, width, rstd.data_ptr(), st), "b2")
    torch.cuda.synchronize()
    assert torch.equal(dwork2, dwork)
    close(dwork[:, :2 * width], ref_in.grad[:, :2 * width], rel=3e-2, what="qkln bwd")


@pytest.mark.parametrize("width,hd,S,B", [(1024, 64, 64, 5), (768, 64, 256, 2), (1024, 64, 77, 3), (128, 32, 40, 4), (2048, 64, 64, 2)])
def test_qkln_head_major(hip, width, hd, S, B):
    """md_qkln_fwd_hm / md_qkln_bwd_hm against the in-place row-major kernels: the SAME bits, re-laid -- normalised q / k as
    [seg][B, H, S, hd], statistics identical, the input untouched; the backward reads dL/dy and y head-major and writes dL/dx into the
    packed rows, leaving the third (v) column block alone."""
    torch.manual_seed(width + S)
    L, st = hip.lib(), hip.stream_ptr()
    rows, ld, H = B * S, 3 * width, width // hd
    buf = bf(torch.randn(rows, ld, device=DEV) * 2 + 0.5)
    work, rstd = buf.clone(), torch.empty(2, rows, device=DEV)
    hip.check(L.md_qkln_fwd(work.data_ptr(), rows, ld, 0, width, 2, width, rstd.data_ptr(), 1e-6, st), "qkln")
    src, out, rstd2 = buf.clone(), torch.zeros(2, B, H, S, hd, device=DEV, dtype=torch.bfloat16), torch.empty(2, rows, device=DEV)
    hip.check(L.md_qkln_fwd_hm(src.data_ptr(), rows, ld, 0, width, 2, width, out.data_ptr(), rows * width, S, hd, rstd2.data_ptr(),
                               1e-6, st), "qkln hm")
    torch.cuda.synchronize()
    assert torch.equal(src, buf), "the head-major forward must not write its input"
    assert torch.equal(rstd2, rstd)

    def to_hm(x):          # [rows, width] -> [B, H, S, hd]
        return x.reshape(B, S, H, hd).permute(0, 2, 1, 3).contiguous()

    assert torch.equal(out[0], to_hm(work[:, :width])) and torch.equal(out[1], to_hm(work[:, width:2 * width]))
    d = bf(torch.randn(rows, ld, device=DEV))
    want = d.clone()
    hip.check(L.md_qkln_bwd(want.data_ptr(), ld, 0, work.data_ptr(), ld, 0, rows, width, 2, width, width, rstd.data_ptr(), st), "b")
    dy = torch.stack([to_hm(d[:, :width]), to_hm(d[:, width:2 * width])])
    got = d.clone()
    got[:, :2 * width] = 7.0                     # must be overwritten, not accumulated into
    hip.check(L.md_qkln_bwd_hm(dy.data_ptr(), rows * width, out.data_ptr(), rows * width, got.data_ptr(), ld, 0, width, rows, width, 2,
                               S, hd, rstd.data_ptr(), st), "b hm")
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # refusals: rows not a multiple of S, head_dim not 32 / 64
    assert L.md_qkln_fwd_hm(src.data_ptr(), rows, ld, 0, width, 2, width, out.data_ptr(), rows * width, S + 1, hd, rstd2.data_ptr(),
                            1e-6, st) == -1
    assert L.md_qkln_fwd_hm(src.data_ptr(), rows, ld, 0, width, 2, width, out.data_ptr(), rows * width, S, 48, rstd2.data_ptr(),
                            1e-6, st) == -1


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H, hd):
    B, Sq, _ = q.shape
    qh = q.view(B, Sq, H, hd).transpose(1, 2)
    kh = k.view(B, -1, H, hd).transpose(1, 2)
    vh = v.view(B, -1, H, hd).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), -1)
    return (att @ vh).transpose(1, 2).reshape(B, Sq, H * hd)


@pytest.mark.parametrize("B,H,Sq,Skv,hd,packed", [(3, 4, 64, 64, 64, True), (2, 12, 256, 256, 64, True),
                                                    (2, 16, 64, 77, 64, False), (2, 3, 77, 77, 64, True),
                                                    (4, 8, 256, 256, 32, True), (2, 4, 40, 77, 32, False),
                                                    (1, 2, 1024, 1024, 64, True), (2, 12, 256, 77, 64, False),
                                                    (3, 2, 96, 200, 64, False), (2, 2, 16, 16, 32, True),
                                                    (2, 3, 1024, 77, 64, False),      # res-512 mixer cross-attention: 4 query blocks x 1 key block
                                                    (1, 2, 600, 300, 64, False),      # ragged 3 x 2 block grid (idle workgroups in a round)
                                                    (1, 2, 300, 700, 32, False)])     # more key blocks than query blocks, head_dim 32
@pytest.mark.parametrize("bwd_split", [0, 2, 3, 4, 5])
@pytest.mark.parametrize("layout", ["packed", "head_major_qk"])
def test_attention(hip, B, H, Sq, Skv, hd, packed, bwd_split, layout):
    """bwd_split 0: the library's choice (ONE fused backward launch per (batch, head) when Sq, Skv <= 256; longer sequences --
    the res-512 mixer, 1024 tokens -- on the streaming pair); 2 / 3 / 4: the fused backward forced to its single-phase (Q, dO, K, V
    in LDS together) or two-phase (half the LDS image; dK / dV in two passes for the 256-row buckets (3) or always (4)) form, all
    three for Sq, Skv <= 256 only; 5: the streaming pair (128-row chunks with register prefetch, up to 8 waves per workgroup) --
    a forced form that does not cover the problem must refuse it (-1) and launch nothing.
    layout "head_major_qk": q, k, dq, dk as [B, H, S, hd] (md_attn_args.hsq / hsk / hsdq / hsdk -- what md_qkln_fwd_hm writes and
    md_qkln_bwd_hm reads), v / o / dO / dv in the packed rows.  All against torch fp32 autograd of the same bf16 inputs
    (utils.py:116-132,177-193)."""
    torch.manual_seed(B * H + Sq + Skv + hd)
    L, st = hip.lib(), hip.stream_ptr()
    hid = H * hd
    hm = layout == "head_major_qk"
    if packed:   # self-attention: [B, S, 3, H, hd]
        qkv = bf(torch.randn(B, Sq, 3 * hid, device=DEV))
        q, k, v = qkv[..., :hid], qkv[..., hid:2 * hid], qkv[..., 2 * hid:]
        ldq = ldk = ldv = 3 * hid
        sq = sk = sv = Sq * 3 * hid
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[..., :hid], dqkv[..., hid:2 * hid], dqkv[..., 2 * hid:]
        lddq = lddk = lddv = 3 * hid
        sdq = sdk = sdv = Sq * 3 * hid
    else:        # cross-attention: q [B,Sq,hid], kv [B,Skv,2,H,hd]
        qb = bf(torch.randn(B, Sq, hid, device=DEV))
        kv = bf(torch.randn(B, Skv, 2 * hid, device=DEV))
        q, k, v = qb, kv[..., :hid], kv[..., hid:]
        ldq, ldk, ldv = hid, 2 * hid, 2 * hid
        sq, sk, sv = Sq * hid, Skv * 2 * hid, Skv * 2 * hid
        dqb = torch.zeros_like(qb)
        dkv = torch.zeros_like(kv)
        dq, dk, dv = dqb, dkv[..., :hid], dkv[..., hid:]
        lddq, lddk, lddv = hid, 2 * hid, 2 * hid
        sdq, sdk, sdv = Sq * hid, Skv * 2 * hid, Skv * 2 * hid
    o = torch.zeros(B, Sq, hid, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, Sq, device=DEV)
    delta = torch.zeros(B, H, Sq, device=DEV)
    do = bf(torch.randn(B, Sq, hid, device=DEV))
    a = hip.AttnArgs(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(),
                     dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), B, H, Sq, Skv,
                     ldq, ldk, ldv, hid, sq, sk, sv, Sq * hid, lddq, lddk, lddv, hid, sdq, sdk, sdv, Sq * hid,
                     1.0 / math.sqrt(hd), hd, bwd_split)
    if hm:       # the same values, q / k re-laid as [B, H, S, hd]; dq / dk come back in that layout
        q_hm = q.reshape(B, Sq, H, hd).permute(0, 2, 1, 3).contiguous()
        k_hm = k.reshape(B, Skv, H, hd).permute(0, 2, 1, 3).contiguous()
        dq_hm, dk_hm = torch.zeros_like(q_hm), torch.zeros_like(k_hm)
        a.q, a.k, a.dq, a.dk = q_hm.data_ptr(), k_hm.data_ptr(), dq_hm.data_ptr(), dk_hm.data_ptr()
        a.ldq = a.ldk = a.lddq = a.lddk = hd
        a.sq = a.sdq = H * Sq * hd
        a.sk = a.sdk = H * Skv * hd
        a.hsq = a.hsdq = Sq * hd
        a.hsk = a.hsdk = Skv * hd
    hip.check(L.md_attn_fwd(byref(a), st), "attn fwd")
    covered = bwd_split in (0, 5) or max(Sq, Skv) <= 256
    rc = L.md_attn_bwd(byref(a), st)
    if not covered:
        torch.cuda.synchronize()
        assert rc == -1, f"forced backward form {bwd_split} must refuse Sq={Sq} Skv={Skv} (rc {rc})"
        got_q, got_k = (dq_hm, dk_hm) if hm else (dq, dk)
        assert float(got_q.abs().max()) == 0.0 and float(got_k.abs().max()) == 0.0, "a refused launch must not write"
        return
    hip.check(rc, "attn bwd")
    a.bwd_split = 1                                   # the round-4 kernel pair was removed with ABI 6: not a valid selector any more
    assert L.md_attn_bwd(byref(a), st) == -1
    if hm:
        dq = dq_hm.permute(0, 2, 1, 3).reshape(B, Sq, hid)
        dk = dk_hm.permute(0, 2, 1, 3).reshape(B, Skv, hid)
    qr = q.float().clone().requires_grad_(True)
    kr = k.float().clone().requires_grad_(True)
    vr = v.float().clone().requires_grad_(True)
    ref = _attn_ref(qr, kr, vr, H, hd)
    ref.backward(do.float())
    torch.cuda.synchronize()
    assert rel_rms(o, ref) < 1e-2, rel_rms(o, ref)
    close(o, ref, rel=3e-2, what="attn out")
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        r = rel_rms(got, want)
        assert r < 2e-2, f"{name} rel-rms {r}"


def test_checksum_u16_is_exact(hip):
    """md_checksum_u16 (the replica-consistency check of the data-parallel step): both sums against exact integer arithmetic in
    numpy, bit-identical across repeated launches, and sensitive to one flipped bit, a sign flip and a swap of two elements."""
    import numpy as np
    torch.manual_seed(3)
    n = 8 * 100003
    x = torch.randn(n, device=DEV).to(torch.bfloat16)
    L, st = hip.lib(), hip.stream_ptr()

    def cs(t):
        out = torch.zeros(2, device=DEV, dtype=torch.int64)
        hip.check(L.md_checksum_u16(t.data_ptr(), t.numel(), out.data_ptr(), st), "md_checksum_u16")
        torch.cuda.synchronize()
        return out.cpu().numpy().astype(np.uint64)

    w = x.view(torch.int16).cpu().numpy().astype(np.uint16).astype(np.uint64)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = ((idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)) | np.uint64(1)
        want = np.array([w.sum(dtype=np.uint64), (w * h).sum(dtype=np.uint64)], dtype=np.uint64)
    got = cs(x)
    assert (got == want).all(), (got, want)
    assert (cs(x) == got).all()
    y = x.clone()
    y.view(torch.int16)[777] ^= 1
    assert (cs(y) != got).any()
    y = x.clone()
    y[5] = -y[5]
    assert (cs(y) != got).any()
    y = x.clone()
    y[[10, 20]] = y[[20, 10]]
    assert cs(y)[0] == got[0] and cs(y)[1] != got[1], "a permutation keeps the plain sum and must change the index-weighted one"


# ------------------------------------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("M,f", [(300, 512), (77 * 3 + 1, 2816), (1031, 776), (65536, 1792), (19, 8)])
def test_swiglu_shapes(hip, M, f):
    """md_swiglu_fwd / md_swiglu_bwd (dit.py:88-89) on ragged row counts (every tail of the 4-rows-per-step walk), widths that
    are not a multiple of the 512-column wave span, padded leading dimensions, and one real XL/2 launch (65,536 x 1792)."""
    torch.manual_seed(M + f)
    L, st = hip.lib(), hip.stream_ptr()
    ldh, lda = 2 * f + 16, f + 8
    h12 = bf(torch.randn(M, ldh, device=DEV))
    a = torch.full((M, lda), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_swiglu_fwd(h12.data_ptr(), ldh, a.data_ptr(), lda, M, f, st), "swiglu")
    h1, h2 = h12[:, :f].float().requires_grad_(True), h12[:, f:2 * f].float().requires_grad_(True)
    ar = F.silu(h1).bfloat16().float() * h2
    da = bf(torch.randn(M, lda, device=DEV))
    ar.backward(da[:, :f].float())
    dh = torch.full((M, ldh), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_swiglu_bwd(da.data_ptr(), lda, h12.data_ptr(), ldh, dh.data_ptr(), ldh, M, f, st), "swiglu bwd")
    torch.cuda.synchronize()
    close(a[:, :f], ar, what="swiglu")
    close(dh[:, :f], h1.grad, what="swiglu bwd d(h1)")
    close(dh[:, f:2 * f], h2.grad, what="swiglu bwd d(h2)")
    assert (a[:, f:] == 7.0).all() and (dh[:, 2 * f:] == 7.0).all(), "padding columns must not be written"


def test_swiglu_gate_act_colsum(hip):
    torch.manual_seed(1)
    L, st = hip.lib(), hip.stream_ptr()
    M, f = 300, 512
    h12 = bf(torch.randn(M, 2 * f, device=DEV))
    a = torch.empty(M, f, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_swiglu_fwd(h12.data_ptr(), 2 * f, a.data_ptr(), f, M, f, st), "swiglu")
    hr = h12.float().requires_grad_(True)
    ar = F.silu(hr[:, :f]) * hr[:, f:]
    da = bf(torch.randn(M, f, device=DEV))
    ar.backward(da.float())
    dh = torch.empty_like(h12)
    hip.check(L.md_swiglu_bwd(da.data_ptr(), f, h12.data_ptr(), 2 * f, dh.data_ptr(), 2 * f, M, f, st), "swiglu bwd")
    torch.cuda.synchronize()
    close(a, ar, what="swiglu")
    close(dh, hr.grad, what="swiglu bwd")
    # gate backward
    rows, C, rps = 384, 768, 64
    B = rows // rps
    dx, br = bf(torch.randn(rows, C, device=DEV)), bf(torch.randn(rows, C, device=DEV))
    mod = bf(torch.randn(B, 6 * C, device=DEV))
    dbr = torch.empty_like(dx)
    dmod = torch.zeros(B, 6 * C, device=DEV)
    hip.check(L.md_gate_bwd(dx.data_ptr(), br.data_ptr(), mod[:, 2 * C:].data_ptr(), 6 * C, dbr.data_ptr(),
                            dmod[:, 2 * C:].data_ptr(), 6 * C, rows, C, rps, 32, st), "gate bwd")
    torch.cuda.synchronize()
    g = mod[:, 2 * C:3 * C].float().repeat_interleave(rps, 0)
    close(dbr, g * dx.float(), what="dbr")
    close(dmod[:, 2 * C:3 * C], (dx.float() * br.float()).view(B, rps, C).sum(1), what="dgate")
    assert dmod[:, :2 * C].abs().max() == 0 and dmod[:, 3 * C:].abs().max() == 0
    # act fwd / bwd
    x = bf(torch.randn(256, 1024, device=DEV))
    y = torch.empty_like(x)
    hip.check(L.md_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), hip.ACT_GELU_TANH, st), "act")
    xr = x.float().requires_grad_(True)
    yr = F.gelu(xr, approximate="tanh")
    dy = torch.randn(256, 1024, device=DEV)
    yr.backward(dy)
    dxo = torch.empty_like(x)
    hip.check(L.md_act_bwd(dy.data_ptr(), x.data_ptr(), dxo.data_ptr(), x.numel(), hip.ACT_GELU_TANH, st), "act bwd")
    torch.cuda.synchronize()
    close(y, yr, what="gelu")
    close(dxo, xr.grad, what="gelu bwd")
    # colsum (bf16 and f32), accumulating
    xs = bf(torch.randn(1000, 520, device=DEV))
    out = torch.ones(520, device=DEV)
    hip.check(L.md_colsum(xs.data_ptr(), 0, 520, out.data_ptr(), 1000, 520, st), "colsum")
    xf = torch.randn(300, 48, device=DEV)
    out2 = torch.zeros(48, device=DEV)
    hip.check(L.md_colsum(xf.data_ptr(), 1, 48, out2.data_ptr(), 300, 48, st), "colsum f32")
    torch.cuda.synchronize()
    close(out, 1 + xs.float().sum(0), rel=1e-3, what="colsum")
    close(out2, xf.sum(0), rel=1e-4, what="colsum f32")
    # casts, mean tokens
    Bc, Lc, Cc = 5, 77, 256
    cap = torch.randn(Bc, Lc, Cc, device=DEV).half()
    drop = torch.tensor([1., 0., 1., 1., 0.], device=DEV)
    yb = torch.empty(Bc * Lc, Cc, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_cast_rows_bf16(cap.data_ptr(), 0, yb.data_ptr(), Bc * Lc, Cc, drop.data_ptr(), Lc, st), "cast rows")
    pooled = torch.empty(Bc, Cc, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_mean_tokens(yb.data_ptr(), pooled.data_ptr(), Bc, Lc, Cc, st), "mean")
    dyf = torch.ones(Bc * Lc, Cc, device=DEV)
    hip.check(L.md_mean_tokens_bwd(pooled.data_ptr(), dyf.data_ptr(), Bc, Lc, Cc, st), "mean bwd")
    torch.cuda.synchronize()
    capr = cap.float() * drop.view(-1, 1, 1)
    close(yb.view(Bc, Lc, Cc), capr, what="cast rows")
    close(pooled, yb.float().view(Bc, Lc, Cc).mean(1), what="mean tokens")
    close(dyf.view(Bc, Lc, Cc), 1 + pooled.float().unsqueeze(1) / Lc, rel=1e-3, what="mean bwd")


# ------------------------------------------------------------------------------------------------ masking / routing
@pytest.mark.parametrize("T,ratio", [(256, 0.75), (1024, 0.75), (64, 0.5)])
def test_get_mask_bit_exact(hip, T, ratio):
    torch.manual_seed(T)
    L, st = hip.lib(), hip.stream_ptr()
    B = 9
    noise = torch.rand(B, T, device=DEV)
    noise[1, 3] = noise[1, T - 1]
    noise[2, :] = 0.25
    noise[3, ::2] = noise[3, 1::2]
    len_keep = int(T * (1 - ratio))
    keep = torch.empty(B * len_keep, dtype=torch.int32, device=DEV)
    restore = torch.empty(B, T, dtype=torch.int32, device=DEV)
    mask = torch.empty(B, T, device=DEV)
    hip.check(L.md_get_mask(noise.data_ptr(), B, T, len_keep, keep.data_ptr(), restore.data_ptr(), mask.data_ptr(), st), "mask")
    torch.cuda.synchronize()
    n = noise.cpu()
    shuffle = torch.argsort(n, dim=1, stable=True)
    r = torch.argsort(shuffle, dim=1, stable=True)
    assert torch.equal(restore.cpu().long(), r)
    assert torch.equal(keep.cpu().long().view(B, len_keep), shuffle[:, :len_keep] + torch.arange(B).view(-1, 1) * T)
    assert torch.equal(mask.cpu(), (r >= len_keep).float())
    # gather / scatter round trip
    C = 128
    x = bf(torch.randn(B * T, C, device=DEV))
    g = torch.empty(B * len_keep, C, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_gather_rows(x.data_ptr(), C, keep.data_ptr(), g.data_ptr(), C, B * len_keep, C, st), "gather")
    back = torch.zeros_like(x)
    hip.check(L.md_scatter_rows(g.data_ptr(), C, keep.data_ptr(), back.data_ptr(), C, B * len_keep, C, st), "scatter")
    torch.cuda.synchronize()
    assert torch.equal(g, x[keep.long()])
    m = mask.view(-1, 1).bool()
    assert torch.equal(back, torch.where(m, torch.zeros_like(x), x))


@pytest.mark.parametrize("B,S,E,d", [(4, 64, 8, 256), (3, 256, 8, 128), (2, 64, 4, 128)])
def test_moe_routing_combine(hip, B, S, E, d):
    torch.manual_seed(S + E)
    L, st = hip.lib(), hip.stream_ptr()
    k = int(2.0 * S / E)
    ld = 8
    M, Bk = B * S, B * k
    logits = torch.zeros(M, ld, device=DEV)
    logits[:, :E] = bf(torch.randn(M, E, device=DEV)).float()
    probs = torch.empty(M, ld, device=DEV)
    rowidx = torch.empty(E, Bk, dtype=torch.int32, device=DEV)
    gval = torch.empty(E, Bk, device=DEV)
    slot = torch.empty(M, E, dtype=torch.int32, device=DEV)
    hip.check(L.md_moe_route(logits.data_ptr(), probs.data_ptr(), ld, B, S, E, k, rowidx.data_ptr(), gval.data_ptr(),
                             slot.data_ptr(), st), "route")
    torch.cuda.synchronize()
    pr = torch.softmax(logits[:, :E], -1).view(B, S, E).requires_grad_(True)
    g, m = torch.topk(pr.permute(0, 2, 1), k, dim=-1)       # [B,E,k]
    close(probs[:, :E], pr.reshape(M, E), rel=1e-5, what="probs")
    # same SETS (ordering inside top-k for exact ties may differ; values are tie-free here) and same order
    got_idx = rowidx.view(E, B, k).permute(1, 0, 2).cpu().long() - (torch.arange(B) * S).view(B, 1, 1)
    assert torch.equal(got_idx, m.cpu())
    close(gval.view(E, B, k).permute(1, 0, 2), g, rel=1e-5, what="gval")
    # combine forward
    h2 = bf(torch.randn(E, Bk, d, device=DEV))
    res = bf(torch.randn(M, d, device=DEV))
    mod = bf(torch.randn(B, 6 * d, device=DEV))
    br = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    out = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_moe_combine(h2.data_ptr(), gval.data_ptr(), slot.data_ptr(), res.data_ptr(), mod[:, 5 * d:].data_ptr(),
                               6 * d, br.data_ptr(), out.data_ptr(), B, S, E, k, d, st), "combine")
    h2r = h2.float().view(E, B, k, d).permute(1, 0, 2, 3).clone().requires_grad_(True)   # [B,E,k,d]
    comb = torch.zeros(B, S, d, device=DEV)
    comb = comb.scatter_add(1, m.reshape(B, E * k, 1).expand(-1, -1, d), (g.unsqueeze(-1) * h2r).reshape(B, E * k, d))
    torch.cuda.synchronize()
    close(br, comb.view(M, d), rel=3e-2, what="combine br")
    gate = mod[:, 5 * d:].float().repeat_interleave(S, 0)
    close(out, res.float() + gate * br.float(), rel=2e-2, what="combine out")
    # combine backward
    dbr = bf(torch.randn(M, d, device=DEV))
    comb.backward(dbr.float().view(B, S, d))
    dh2 = torch.empty_like(h2)
    dg = torch.empty(E, Bk, device=DEV)
    hip.check(L.md_moe_combine_bwd(dbr.data_ptr(), h2.data_ptr(), rowidx.data_ptr(), gval.data_ptr(), dh2.data_ptr(),
                                   dg.data_ptr(), E * Bk, d, st), "combine bwd")
    torch.cuda.synchronize()
    close(dh2.view(E, B, k, d).permute(1, 0, 2, 3), h2r.grad, rel=2e-2, what="dh2")
    # dispatch backward + softmax backward
    dxin = bf(torch.randn(E, Bk, d, device=DEV))
    dx = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    dlog = torch.empty(M, ld, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_moe_dispatch_bwd(dxin.data_ptr(), slot.data_ptr(), dx.data_ptr(), probs.data_ptr(), ld, dg.data_ptr(),
                                    dlog.data_ptr(), ld, B, S, E, k, d, st), "dispatch bwd")
    torch.cuda.synchronize()
    ref_dx = torch.zeros(B, S, d, device=DEV).scatter_add(
        1, m.reshape(B, E * k, 1).expand(-1, -1, d), dxin.float().view(E, B, k, d).permute(1, 0, 2, 3).reshape(B, E * k, d))
    close(dx, ref_dx.view(M, d), rel=2e-2, what="dispatch bwd")
    # dlogits through softmax: compare against autograd of pr wrt logits
    lg = logits[:, :E].clone().requires_grad_(True)
    pr2 = torch.softmax(lg, -1).view(B, S, E)
    g2, m2 = torch.topk(pr2.permute(0, 2, 1), k, dim=-1)
    (g2 * dg.view(E, B, k).permute(1, 0, 2)).sum().backward()
    close(dlog[:, :E], lg.grad, rel=3e-2, what="dlogits")
    if E < ld:
        assert dlog[:, E:].float().abs().max() == 0


# ------------------------------------------------------------------------------------------------ EDM front/back end
@pytest.mark.parametrize("masked", [True, False])
def test_edm_patchify_loss(hip, masked):
    torch.manual_seed(3)
    L, st = hip.lib(), hip.stream_ptr()
    B, C, H, W, p = 6, 4, 32, 32, 2
    T = (H // p) * (W // p)
    x0 = torch.randn(B, C, H, W, device=DEV) * 0.8
    eps = torch.randn(B, C, H, W, device=DEV)
    rnd = torch.randn(B, device=DEV)
    xn, sigma, cin, cnoise = torch.empty_like(x0), torch.empty(B, device=DEV), torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    hip.check(L.md_edm_prepare(x0.data_ptr(), eps.data_ptr(), rnd.data_ptr(), xn.data_ptr(), sigma.data_ptr(), cin.data_ptr(),
                               cnoise.data_ptr(), B, C * H * W, -0.6, 1.2, 0.9, st), "prepare")
    patches = torch.empty(B * T, C * p * p, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_patchify(xn.data_ptr(), cin.data_ptr(), patches.data_ptr(), B, C, H, W, p, st), "patchify")
    torch.cuda.synchronize()
    s = (rnd * 1.2 - 0.6).exp()
    close(sigma, s, rel=1e-5, what="sigma")
    close(xn, x0 + eps * s.view(-1, 1, 1, 1), rel=1e-5, what="xn")
    close(cnoise, s.log() / 4, rel=1e-4, what="cnoise")
    c_in = 1 / (0.81 + s * s).sqrt()
    ref_p = F.unfold(xn * c_in.view(-1, 1, 1, 1), p, stride=p).transpose(1, 2).reshape(B * T, C * p * p)
    close(patches, ref_p, what="patchify")
    # timestep embedding
    te = torch.empty(B, 512, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_timestep_embed(cnoise.data_ptr(), te.data_ptr(), B, 512, st), "temb")
    torch.cuda.synchronize()
    fr = torch.exp(-math.log(10000) * torch.arange(256, device=DEV, dtype=torch.float32) / 256)
    ar = cnoise[:, None] * fr[None]
    close(te, torch.cat([ar.cos(), ar.sin()], -1), what="timestep emb")
    # unpatchify + loss
    Tk = T // 4 if masked else T
    if masked:
        noise = torch.rand(B, T, device=DEV)
        keep = torch.empty(B * Tk, dtype=torch.int32, device=DEV)
        restore = torch.empty(B, T, dtype=torch.int32, device=DEV)
        mask = torch.empty(B, T, device=DEV)
        hip.check(L.md_get_mask(noise.data_ptr(), B, T, Tk, keep.data_ptr(), restore.data_ptr(), mask.data_ptr(), st), "m")
    tok = bf(torch.randn(B * Tk, C * p * p, device=DEV))
    img = torch.empty(B, C, H, W, device=DEV)
    hip.check(L.md_unpatchify(tok.data_ptr(), restore.data_ptr() if masked else None, Tk, None, img.data_ptr(), B, C, H, W, p,
                              st), "unpatchify")
    lps, lmean = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    dtok = torch.empty(B * Tk, C * p * p, device=DEV)
    hip.check(L.md_edm_loss(tok.data_ptr(), keep.data_ptr() if masked else None, xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(),
                            lps.data_ptr(), lmean.data_ptr(), dtok.data_ptr(), B, Tk, C, H, W, p, 0.9, st), "loss")
    torch.cuda.synchronize()
    tr = tok.float().clone().requires_grad_(True)
    xt = tr.view(B, Tk, -1)
    if masked:
        xt = torch.cat([xt, torch.zeros(B, T - Tk, C * p * p, device=DEV)], 1)
        xt = torch.gather(xt, 1, restore.long().unsqueeze(-1).expand(-1, -1, C * p * p))
    g = H // p
    Fx = xt.view(B, g, g, p, p, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)
    close(img, Fx, rel=1e-6, what="unpatchify")
    sg = s.view(-1, 1, 1, 1)
    D = 0.81 / (sg ** 2 + 0.81) * xn + sg * 0.9 / (sg ** 2 + 0.81).sqrt() * Fx
    loss = (sg ** 2 + 0.81) / (sg * 0.9) ** 2 * (D - x0) ** 2
    if masked:
        lp = F.avg_pool2d(loss.mean(1), p).flatten(1)
        um = 1 - mask
        lp = (lp * um).sum(1) / um.sum(1)
    else:
        lp = loss.flatten(1).mean(1)
    lp.mean().backward()
    close(lps, lp, rel=1e-4, what="loss per sample")
    close(lmean, lp.mean().view(1), rel=1e-4, what="loss mean")
    close(dtok, tr.grad, rel=1e-3, what="dtok")
    # ---- the training-step forms: fp16 latents in (fp32 copy out), bf16 dL/dF pre-multiplied by the microbatch weight, loss
    # accumulated on the device; and the batch mean is a fixed-order sum (bit-identical from run to run)
    x0h = x0.half()
    xn2, x0f = torch.empty_like(x0), torch.empty_like(x0)
    hip.check(L.md_edm_prepare_f16(x0h.data_ptr(), eps.data_ptr(), rnd.data_ptr(), xn2.data_ptr(), x0f.data_ptr(), sigma.data_ptr(),
                                   cin.data_ptr(), cnoise.data_ptr(), B, C * H * W, -0.6, 1.2, 0.9, st), "prepare_f16")
    torch.cuda.synchronize()
    assert torch.equal(x0f, x0h.float())
    close(xn2, x0h.float() + eps * s.view(-1, 1, 1, 1), rel=1e-6, what="xn from fp16 latents")
    w = 0.375
    dtb = torch.empty(B * Tk, C * p * p, device=DEV, dtype=torch.bfloat16)
    lps2, lmean2, acc = torch.empty(B, device=DEV), torch.empty(1, device=DEV), torch.full((1,), 2.0, device=DEV)
    hip.check(L.md_edm_loss_train(tok.data_ptr(), keep.data_ptr() if masked else None, xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(),
                                  lps2.data_ptr(), lmean2.data_ptr(), dtb.data_ptr(), w, acc.data_ptr(), w, B, Tk, C, H, W, p, 0.9, st), "loss_train")
    torch.cuda.synchronize()
    assert torch.equal(lps2, lps) and torch.equal(lmean2, lmean)
    assert abs(acc.item() - (2.0 + w * lmean.item())) <= 1e-6 * abs(acc.item())
    close(dtb, (dtok * w).bfloat16(), rel=1e-6, what="bf16 dtok pre-scaled by the microbatch weight")
    for _ in range(3):
        lm = torch.empty(1, device=DEV)
        hip.check(L.md_edm_loss(tok.data_ptr(), keep.data_ptr() if masked else None, xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(),
                                lps2.data_ptr(), lm.data_ptr(), None, B, Tk, C, H, W, p, 0.9, st), "loss")
        torch.cuda.synchronize()
        assert torch.equal(lm, lmean), "the batch-mean loss must not depend on arrival order"


def test_gemm_operand_lists(hip):
    """md_gemm_args.A_list / B_list (pp256, fp32 slices): every (batch, split) item reads its own operand pair; with
    md_splitk_reduce the launch computes sum_l A_l B_l -- the grouped adaLN condition-vector gradient (28 layers x 2 k-parts)."""
    torch.manual_seed(9)
    L, st = hip.lib(), hip.stream_ptr()
    Bm, D, N, G, parts = 256, 1024, 1536, 5, 2
    kspan = N // parts
    dm = bf(torch.randn(G, Bm, N, device=DEV))
    Ws = [bf(torch.randn(N, D, device=DEV) / math.sqrt(N)) for _ in range(G)]            # separate allocations on purpose
    al = torch.tensor([dm[i].data_ptr() + 2 * c * kspan for i in range(G) for c in range(parts)], dtype=torch.int64).to(DEV)
    bl = torch.tensor([Ws[i].data_ptr() + 2 * c * kspan * D for i in range(G) for c in range(parts)], dtype=torch.int64).to(DEV)
    ks = G * parts
    ws = torch.empty(ks, Bm, D, device=DEV)
    chosen = []
    hip.gemm(A=dm, B=Ws[0], C=ws, M=Bm, N=D, K=kspan * ks, lda=N, ldb=D, ldc=D, sC=ks * Bm * D, sSplit=Bm * D, ksplit=ks,
             a_kcontig=1, b_kcontig=0, mode=hip.EPI_STORE_F32, A_list=al, B_list=bl, chosen=chosen)
    out = torch.full((Bm, D), 1.0, device=DEV)
    hip.check(L.md_splitk_reduce(ws.data_ptr(), out.data_ptr(), Bm, D, D, 0, ks, 1, 1, st), "reduce")
    torch.cuda.synchronize()
    assert chosen == [hip.GEMM_PP256]
    ref = 1.0 + sum(dm[i].float() @ Ws[i].float() for i in range(G))
    close(out, ref, rel=2e-3, what="sum over operand-list items")
    # ---- list_segments: ONE item walks G operand pairs and keeps the sum in its accumulators (K-concatenation), optionally
    # split over groups of segments
    Mc, Kk, G2 = 640, 256, 6
    dk = bf(torch.randn(G2, Mc, Kk, device=DEV))
    Wk = [bf(torch.randn(Kk, D, device=DEV) / math.sqrt(Kk * G2)) for _ in range(G2)]
    al2 = torch.tensor([dk[i].data_ptr() for i in range(G2)], dtype=torch.int64).to(DEV)
    bl2 = torch.tensor([w_.data_ptr() for w_ in Wk], dtype=torch.int64).to(DEV)
    ref2 = sum(dk[i].float() @ Wk[i].float() for i in range(G2))
    for ks2 in (1, 2, 3):
        ws2 = torch.empty(ks2, Mc, D, device=DEV)
        hip.gemm(A=dk, B=Wk[0], C=ws2, M=Mc, N=D, K=Kk * G2, lda=Kk, ldb=D, ldc=D, sC=ks2 * Mc * D, sSplit=Mc * D, ksplit=ks2,
                 a_kcontig=1, b_kcontig=0, mode=hip.EPI_STORE_F32, A_list=al2, B_list=bl2, list_segments=G2 // ks2)
        out2 = torch.zeros(Mc, D, device=DEV)
        hip.check(L.md_splitk_reduce(ws2.data_ptr(), out2.data_ptr(), Mc, D, D, 0, ks2, 1, 1, st), "reduce")
        torch.cuda.synchronize()
        close(out2, ref2, rel=2e-3, what=f"K-concatenated operand lists, ksplit {ks2}")
    # lists on a kernel that is not built for them are refused as a bad argument, not silently ignored
    C = torch.empty(Bm, D, device=DEV, dtype=torch.bfloat16)
    assert hip.gemm(A=dm, B=Ws[0], C=C, M=Bm, N=D, K=kspan, lda=N, ldb=D, ldc=D, a_kcontig=1, b_kcontig=0, A_list=al, B_list=bl,
                    expect=None) == -1


@pytest.mark.parametrize("ks", [1, 2, 4])
def test_gemm_grouped_problems(hip, ks):
    """md_gemm_args.problems: several weight gradients that contract over the same tokens as ONE pp256 launch (tiles of all
    problems share the workgroups), fp32 slices laid out like the gradient tensors (a gap between two of them is left alone),
    one md_splitk_reduce_flat per contiguous run.  Shapes: qkv / proj / q_linear-like plus ragged ones (rows not a multiple of
    256, columns only a multiple of 8)."""
    import ctypes
    torch.manual_seed(11 + ks)
    L, st = hip.lib(), hip.stream_ptr()
    T = 1024                                               # tokens (the shared contraction)
    shapes = [(1920, 1024), (1024, 640), (1024, 1024), (200, 72), (520, 264)]          # (out rows = N_lin, out cols = K_lin)
    gap_after = 1                                          # a foreign tensor (kv_linear in a real block) between problems 1 and 2
    sizes = [m * n for m, n in shapes]
    offs, o = [], 0
    for i, sz in enumerate(sizes):
        offs.append(o)
        o += (sz + 63) // 64 * 64
        if i == gap_after:
            o += 4096
    span = o
    G = torch.full((span,), 0.5, device=DEV)               # the "gradient accumulators": reduced into with accumulate = 1
    dys = [bf(torch.randn(T, m, device=DEV)) for m, _ in shapes]
    xs = [bf(torch.randn(T, n, device=DEV) / math.sqrt(T)) for _, n in shapes]
    probs = (hip.GemmProblem * len(shapes))()
    for i, (m, n) in enumerate(shapes):
        probs[i] = hip.GemmProblem(dys[i].data_ptr(), xs[i].data_ptr(), m, n, m, n, offs[i])
    ws = torch.full((ks, span), float("nan"), device=DEV)
    a = hip.GemmArgs()
    for k, v in dict(A=dys[0].data_ptr(), B=xs[0].data_ptr(), C=ws.data_ptr(), M=shapes[0][0], N=shapes[0][1], K=T, lda=shapes[0][0],
                     ldb=shapes[0][1], ldc=shapes[0][1], sSplit=span, batch=1, ksplit=ks, a_kcontig=0, b_kcontig=0,
                     mode=hip.EPI_STORE_F32, act=0, alpha=1.0, problems=ctypes.addressof(probs), n_problems=len(shapes)).items():
        setattr(a, k, v)
    hip.check(L.md_gemm_bf16(ctypes.byref(a), st), "grouped gemm")
    runs = [(offs[0], offs[1] + sizes[1]), (offs[2], offs[4] + sizes[4])]
    for lo, hi in runs:
        n = ((hi - lo) + 3) // 4 * 4
        hip.check(L.md_splitk_reduce_flat(ws.data_ptr() + 4 * lo, G.data_ptr() + 4 * lo, n, span, ks, 1, st), "flat reduce")
    torch.cuda.synchronize()
    for i, (m, n) in enumerate(shapes):
        ref = 0.5 + dys[i].float().t() @ xs[i].float()
        close(G[offs[i]:offs[i] + m * n].view(m, n), ref, rel=2e-3, what=f"grouped problem {i} {m}x{n}, ksplit {ks}")
    gap = G[offs[1] + sizes[1]:offs[2]]
    pad_end = (sizes[1] + 63) // 64 * 64 - sizes[1]
    assert torch.all(gap[pad_end:] == 0.5), "the gap between two runs must not be touched"
    # a grouped launch on anything but the K-strided x K-strided fp32-slice kernel is a bad argument
    a.a_kcontig = 1
    assert L.md_gemm_bf16(ctypes.byref(a), st) == -1


@pytest.mark.parametrize("ks", [1, 2, 4])
@pytest.mark.parametrize("variant", ["pp256", "w4", "auto"])
def test_gemm_grouped_problems_whole_tiles(hip, variant, ks):
    """The grouped weight-gradient launch on shapes of whole 256 x 256 tiles (what a DiT block's group is), forced through pp256 and
    through the 4-wave kernel (round 6: both operands K-strided through transposing LDS reads, fp32 slices stored 16 bytes per lane),
    and by the library's own choice (enough tiles: w4)."""
    import ctypes
    torch.manual_seed(17 + ks)
    L, st = hip.lib(), hip.stream_ptr()
    T = 2048
    shapes = [(1024, 768), (768, 1024), (512, 256), (256, 2816)]
    sizes = [m * n for m, n in shapes]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    span = sum(sizes)
    dys = [bf(torch.randn(T, m, device=DEV)) for m, _ in shapes]
    xs = [bf(torch.randn(T, n, device=DEV) / math.sqrt(T)) for _, n in shapes]
    probs = (hip.GemmProblem * len(shapes))()
    for i, (m, n) in enumerate(shapes):
        probs[i] = hip.GemmProblem(dys[i].data_ptr(), xs[i].data_ptr(), m, n, m, n, offs[i])
    ws = torch.full((ks, span), float("nan"), device=DEV)
    chosen = ctypes.c_int32(-1)
    a = hip.GemmArgs()
    for k, v in dict(A=dys[0].data_ptr(), B=xs[0].data_ptr(), C=ws.data_ptr(), M=shapes[0][0], N=shapes[0][1], K=T, lda=shapes[0][0],
                     ldb=shapes[0][1], ldc=shapes[0][1], sSplit=span, batch=1, ksplit=ks, a_kcontig=0, b_kcontig=0,
                     mode=hip.EPI_STORE_F32, act=0, alpha=1.0, problems=ctypes.addressof(probs), n_problems=len(shapes),
                     variant=hip.GEMM_VARIANT_NAMES[variant], chosen_variant=ctypes.addressof(chosen)).items():
        setattr(a, k, v)
    hip.check(L.md_gemm_bf16(ctypes.byref(a), st), "grouped gemm")
    if variant != "auto":
        assert chosen.value == hip.GEMM_VARIANT_NAMES[variant]
    G = torch.full((span,), 0.25, device=DEV)
    hip.check(L.md_splitk_reduce_flat(ws.data_ptr(), G.data_ptr(), span, span, ks, 1, st), "flat reduce")
    torch.cuda.synchronize()
    assert torch.isfinite(ws).all()
    for i, (m, n) in enumerate(shapes):
        ref = 0.25 + dys[i].float().t() @ xs[i].float()
        close(G[offs[i]:offs[i] + m * n].view(m, n), ref, rel=2e-3, what=f"{variant}: grouped problem {i} {m}x{n}, ksplit {ks}")


# ------------------------------------------------------------------------------------------------ optimiser
def test_adamw_clip(hip):
    """clip_grad_norm_ + torch.optim.AdamW (train.py:39-43,85-86) vs md_sumsq / md_sumsq_finish / md_adamw_step, with the
    gradient taken from the fp32 accumulators and from a bf16 exchange buffer, and the EMA of the weights
    (configs/res_512_pretrain.yaml:4-9) folded in: ema <- weights on its first batch, s * ema + (1 - s) * weights after."""
    torch.manual_seed(4)
    L, st = hip.lib(), hip.stream_ptr()
    n = 4096 * 33
    for g_bf16 in (False, True):
        p = torch.randn(n, device=DEV)
        pref = torch.nn.Parameter(p.clone())
        opt = torch.optim.AdamW([pref], lr=2.4e-4, weight_decay=0.1, eps=1e-8, betas=(0.9, 0.999))
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
        ema = torch.zeros(n, device=DEV)
        ema_ref = None
        ss = torch.zeros(1, device=DEV)
        part = torch.zeros(hip.SUMSQ_PARTIALS, device=DEV)
        for step in range(1, 5):
            g = torch.randn(n, device=DEV) * (0.01 if step == 3 else 1.0)   # step 3: below the clip threshold
            if g_bf16:
                g = bf(g).float()
            pref.grad = g.clone()
            torch.nn.utils.clip_grad_norm_([pref], 0.25 if step != 3 else 1e9)
            opt.step()
            gw = g.clone()
            gb = bf(g) if g_bf16 else None
            src = gb if g_bf16 else gw
            hip.check(L.md_sumsq(src.data_ptr(), 1 if g_bf16 else 0, n, part.data_ptr(), st), "sumsq")
            hip.check(L.md_sumsq_finish(part.data_ptr(), hip.SUMSQ_PARTIALS, ss.data_ptr(), st), "sumsq_finish")
            ema_mode = 0 if step == 1 else (1 if step == 2 else 2)
            a = hip.AdamWArgs(p.data_ptr(), gw.data_ptr(), m.data_ptr(), v.data_ptr(), shadow.data_ptr(), ss.data_ptr(),
                              gb.data_ptr() if g_bf16 else None, ema.data_ptr(), n,
                              2.4e-4, 0.9, 0.999, 1e-8, 0.1, 1 - 0.9 ** step, 1 - 0.999 ** step,
                              0.25 if step != 3 else 1e9, 1.0, 0.99, 1, ema_mode)
            hip.check(L.md_adamw_step(byref(a), st), "adamw")
            torch.cuda.synchronize()
            assert abs(ss.item() - (g.double() ** 2).sum().item()) < 1e-5 * (g.double() ** 2).sum().item()
            assert torch.allclose(p, pref.detach(), rtol=1e-5, atol=1e-6), (p - pref.detach()).abs().max()
            assert gw.abs().max() == 0
            assert torch.equal(shadow, p.to(torch.bfloat16))
            if ema_mode == 1:
                ema_ref = pref.detach().clone()
            elif ema_mode == 2:
                ema_ref = 0.99 * ema_ref + 0.01 * pref.detach()
            if ema_ref is not None:
                assert torch.allclose(ema, ema_ref, rtol=1e-5, atol=1e-6)
    # determinism of the norm: bit-identical partial sums run to run
    g = torch.randn(1 << 22, device=DEV)
    outs = []
    for _ in range(3):
        hip.check(L.md_sumsq(g.data_ptr(), 0, g.numel(), part.data_ptr(), st), "sumsq")
        hip.check(L.md_sumsq_finish(part.data_ptr(), hip.SUMSQ_PARTIALS, ss.data_ptr(), st), "sumsq_finish")
        torch.cuda.synchronize()
        outs.append(ss.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


# ------------------------------------------------------------------------------------------------ MoE layer, routing injected
@pytest.mark.parametrize("variant", ["auto", "pp256"])
@pytest.mark.parametrize("B,S,d,f", [(16, 64, 1024, 3840), (4, 256, 768, 3072)])
def test_moe_layer_with_oracle_routing(hip, variant, B, S, d, f):
    """SURVEY.md section 8c golden (ii): the expert-choice MoE layer (dit.py:126-143) at XL/2 geometry (backbone block 25:
    dim 1024, hidden 3840, 64 kept tokens; mixer: dim 768, hidden 3072, 256 tokens; 8 experts, capacity 2) with the
    ORACLE's top-k indices injected, so routing flips cannot hide (or be blamed for) arithmetic error: gather -> grouped
    fc1 (GELU-erf) -> grouped fc2 -> gate-weighted combine through the HIP kernels vs oracle.ec_moe in fp32 on the same
    bf16-rounded weights and inputs.  With identical routing the only difference is bf16 storage of xin / h / h2."""
    from oracle import microdit_ref as orc
    torch.manual_seed(B + S)
    L, st = hip.lib(), hip.stream_ptr()
    E, k = 8, int(2.0 * S / 8)
    M, Bk = B * S, B * k
    x = bf(torch.randn(B, S, d))
    sd = {"m.gate.weight": torch.randn(E, d) * (2.0 / math.sqrt(d)), "m.w1": bf(torch.randn(E, d, f) / math.sqrt(d)).float(),
          "m.w2": bf(torch.randn(E, f, d) / math.sqrt(f)).float()}
    ref, m_idx, g = orc.ec_moe(sd, "m", x.float(), E, 2.0, return_routing=True)          # [B,S,d], [B,E,k], [B,E,k]
    # ---- routing tables in the HIP kernels' layout, from the oracle's indices
    rowidx = (m_idx + (torch.arange(B) * S).view(B, 1, 1)).permute(1, 0, 2).reshape(E, Bk).to(torch.int32).to(DEV)
    gval = g.permute(1, 0, 2).reshape(E, Bk).contiguous().to(DEV)
    slot = torch.full((B, S, E), -1, dtype=torch.int32)
    for e in range(E):
        slot[:, :, e].scatter_(1, m_idx[:, e, :], torch.arange(k, dtype=torch.int32).view(1, k).expand(B, k))
    slot = slot.view(M, E).to(DEV)
    xg = x.view(M, d).to(DEV)
    xin = torch.empty(E, Bk, d, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_gather_rows(xg.data_ptr(), d, rowidx.data_ptr(), xin.data_ptr(), d, E * Bk, d, st), "gather")
    w1, w2 = bf(sd["m.w1"]).to(DEV), bf(sd["m.w2"]).to(DEV)
    hact = torch.empty(E, Bk, f, device=DEV, dtype=torch.bfloat16)
    hpre = torch.empty_like(hact)
    v = hip.GEMM_VARIANT_NAMES[variant]

    def gemm(**kw):
        rc = hip.gemm(variant=v, expect=None, **kw)
        if rc == hip.NOT_ELIGIBLE:
            rc = hip.gemm(variant=hip.GEMM_AUTO, expect=None, **kw)
        hip.check(rc, "gemm")
    gemm(A=xin, B=w1, C=hact, C2=hpre, M=Bk, N=f, K=d, lda=d, ldb=f, ldc=f, ldc2=f, sA=Bk * d, sB=d * f, sC=Bk * f, sC2=Bk * f,
         batch=E, a_kcontig=1, b_kcontig=0, act=hip.ACT_GELU_ERF)
    h2 = torch.empty(E, Bk, d, device=DEV, dtype=torch.bfloat16)
    gemm(A=hact, B=w2, C=h2, M=Bk, N=d, K=f, lda=f, ldb=d, ldc=d, sA=Bk * f, sB=f * d, sC=Bk * d, batch=E, a_kcontig=1, b_kcontig=0)
    res = torch.zeros(M, d, device=DEV, dtype=torch.bfloat16)
    gate = torch.ones(B, 6 * d, device=DEV, dtype=torch.bfloat16)
    br = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    out = torch.empty(M, d, device=DEV, dtype=torch.bfloat16)
    hip.check(L.md_moe_combine(h2.data_ptr(), gval.data_ptr(), slot.data_ptr(), res.data_ptr(), gate[:, 5 * d:].data_ptr(), 6 * d,
                               br.data_ptr(), out.data_ptr(), B, S, E, k, d, st), "combine")
    torch.cuda.synchronize()
    rr = rel_rms(br.cpu().view(B, S, d), ref)
    assert rr <= 0.01, f"MoE layer with injected routing: rel-RMS {rr:.4f} vs fp32 oracle (bf16 storage only: expected ~0.004)"
    close(out, br, rel=1e-2, what="res 0 + gate 1")


def test_adamw_step_ranges(hip):
    """md_adamw_step_ranges (the sharded optimiser step as ONE launch): chunks of the flat buffers named by a range table, the bf16
    gradient and the bf16 weight output packed back to back -- against md_adamw_step chunk by chunk."""
    import ctypes
    torch.manual_seed(12)
    L, st = hip.lib(), hip.stream_ptr()
    n = 64 * 700
    ranges = [(64 * 3, 64 * 10), (64 * 40, 64 * 100), (64 * 300, 64 * 7), (64 * 600, 64 * 100)]
    packed = sum(c for _, c in ranges)
    p0, m0, v0 = torch.randn(n, device=DEV), torch.randn(n, device=DEV) * 0.1, torch.rand(n, device=DEV) * 0.01
    gpk = bf(torch.randn(packed, device=DEV))
    ss = torch.full((1,), float((gpk.float() ** 2).sum()), device=DEV)
    res = {}
    for mode in ("ranges", "chunks"):
        p, m, v, ema = p0.clone(), m0.clone(), v0.clone(), p0.clone() * 0.5
        g = torch.zeros(n, device=DEV)
        spk = torch.zeros(packed, device=DEV, dtype=torch.bfloat16)

        def args(off, cnt, gptr, sptr):
            return hip.AdamWArgs(p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, sptr, ss.data_ptr(),
                                 gptr, ema.data_ptr() + 4 * off, cnt, 1e-3, 0.9, 0.999, 1e-8, 0.1, 1 - 0.9 ** 3, 1 - 0.999 ** 3, 0.25, 0.125, 0.99, 0, 2)
        if mode == "ranges":
            off = (ctypes.c_int64 * len(ranges))(*[o for o, _ in ranges])
            cnt = (ctypes.c_int64 * len(ranges))(*[c for _, c in ranges])
            a = args(0, 0, gpk.data_ptr(), spk.data_ptr())
            hip.check(L.md_adamw_step_ranges(ctypes.byref(a), off, cnt, len(ranges), st), "ranges")
        else:
            o2 = 0
            for o, c in ranges:
                a = args(o, c, gpk.data_ptr() + 2 * o2, spk.data_ptr() + 2 * o2)
                hip.check(L.md_adamw_step(ctypes.byref(a), st), "chunk")
                o2 += c
        torch.cuda.synchronize()
        res[mode] = (p, m, v, ema, spk)
    for a_, b_ in zip(res["ranges"], res["chunks"]):
        assert torch.equal(a_, b_)
    touched = torch.zeros(n, dtype=torch.bool, device=DEV)
    for o, c in ranges:
        touched[o:o + c] = True
    assert torch.equal(res["ranges"][0][~touched], p0[~touched]) and not torch.equal(res["ranges"][0][touched], p0[touched])
