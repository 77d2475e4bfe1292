"""Every kernel behind md_gemm_bf16, forced through md_gemm_args.variant, on the shapes the MicroDiT-XL/2 step really
launches (microbatch 1024: 65,536 backbone tokens, 78,848 caption tokens) and on ragged shapes, against torch fp32 matmul
of the same bf16 operands.  Tolerance: fp32-accumulated bf16 products, outputs rounded to bf16:
|err| <= 2e-2 * max|ref| (+1e-3); fp32 outputs (split-K slices + md_splitk_reduce) 2e-3 * max|ref|.

This closes the round-1 gap: 31 % of the benchmarked step ran on instantiations no test executed.
A variant that cannot run a problem (pp256: K / ksplit not a multiple of 128, ...) must REFUSE it (-1), never fall back.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

VARIANTS = ["reg128", "dma128", "paced256", "pp256", "w4"]
dev = "cuda"


def _operand(rows, k, kcontig, scale=1.0):
    t = (torch.randn(rows, k, device=dev) * scale).to(torch.bfloat16)
    if kcontig:
        return t, t, k
    return t, t.t().contiguous(), rows


def _run(hip, variant, **kw):
    rc = hip.gemm(variant=hip.GEMM_VARIANT_NAMES[variant], expect=None, **kw)
    if rc == hip.NOT_ELIGIBLE:
        return False
    hip.check(rc, "md_gemm_bf16")
    torch.cuda.synchronize()
    return True


def _close(out, ref, rel=2e-2, what=""):
    err = (out.float() - ref).abs().max().item()
    lim = rel * ref.abs().max().item() + 1e-3
    assert err <= lim, f"{what}: max err {err} > {lim}"


# (M, N, K, akc, bkc): forward NT, dgrad NN — the heaviest XL/2 activation GEMMs (profiles/r1_gemm_final_shape_tables.txt)
ACT_SHAPES = [
    (65536, 1024, 1024, 1, 1), (65536, 1024, 1024, 1, 0), (78848, 2048, 1024, 1, 1), (78848, 1024, 2048, 1, 0),
    (65536, 3072, 1024, 1, 1), (65536, 1024, 2816, 1, 0), (32768, 768, 768, 1, 1),
    (1024 + 72, 512 + 40, 1152, 1, 1), (1024 + 72, 512 + 40, 1152, 1, 0),       # ragged M / N on every tile size
]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("M,N,K,akc,bkc", ACT_SHAPES)
def test_plain_store(hip, variant, M, N, K, akc, bkc):
    torch.manual_seed(M + N + K + akc + 2 * bkc)
    A, As, lda = _operand(M, K, akc)
    B, Bs, ldb = _operand(N, K, bkc, 0.05)
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    if not _run(hip, variant, A=As, B=Bs, C=C, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, a_kcontig=akc, b_kcontig=bkc):
        pytest.skip(f"{variant} refuses this problem")
    _close(C, A.float() @ B.float().t(), what=f"{variant} {M}x{N}x{K}")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("M,N,K,bkc,rps", [(65536, 1024, 2816, 1, 64), (65536, 768, 768, 1, 256), (16384 + 72, 1024 + 40, 1152, 1, 64),
                                           (65536, 1024, 1024, 0, 0)])
def test_gated_residual(hip, variant, M, N, K, bkc, rps):
    """proj / w3 epilogue (dit.py:236,238): out = res + gate[sample] * bf16(A W^T + b), raw copy in C2; rps = 0: plain
    residual accumulate of a dgrad (out aliases res)."""
    torch.manual_seed(7 + M + K)
    A, As, lda = _operand(M, K, 1)
    B, Bs, ldb = _operand(N, K, bkc, 0.05)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    raw = A.float() @ B.float().t()
    if rps:
        ns = (M + rps - 1) // rps
        gate = torch.randn(ns, N, device=dev).to(torch.bfloat16)
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        C2 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        ok = _run(hip, variant, A=As, B=Bs, C=out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, a_kcontig=1, b_kcontig=bkc,
                  mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=rps, C2=C2, ldc2=N)
        if not ok:
            pytest.skip(f"{variant} refuses this problem")
        ref = res.float() + gate.float().repeat_interleave(rps, 0)[:M] * raw.to(torch.bfloat16).float()
        _close(C2, raw, what="raw copy")
    else:
        out = res.clone()
        ok = _run(hip, variant, A=As, B=Bs, C=out, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, a_kcontig=1, b_kcontig=bkc,
                  mode=hip.EPI_RESIDUAL, res=out, ldr=N)
        if not ok:
            pytest.skip(f"{variant} refuses this problem")
        ref = res.float() + raw.to(torch.bfloat16).float()
    _close(out, ref, what=f"{variant} residual")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("Bk,d,f", [(16384, 1024, 3840), (65536, 768, 3072), (4096 + 24, 256, 640)])
def test_moe_grouped(hip, variant, Bk, d, f):
    """The four grouped launches of one expert-choice MoE layer (dit.py:131-142; 8 experts, [E, d, f] / [E, f, d] weights):
    fc1 (GELU-erf + raw copy), fc2, dgrad of fc2 through GELU' (DACT), dgrad of fc1."""
    torch.manual_seed(3 + Bk + f)
    E = 8
    X = torch.randn(E, Bk, d, device=dev).to(torch.bfloat16)
    W1 = (torch.randn(E, d, f, device=dev) * 0.03).to(torch.bfloat16)
    W2 = (torch.randn(E, f, d, device=dev) * 0.03).to(torch.bfloat16)
    H = torch.empty(E, Bk, f, device=dev, dtype=torch.bfloat16)
    Hp = torch.empty_like(H)
    ok = _run(hip, variant, A=X, B=W1, C=H, C2=Hp, M=Bk, N=f, K=d, lda=d, ldb=f, ldc=f, ldc2=f, sA=Bk * d, sB=d * f, sC=Bk * f,
              sC2=Bk * f, batch=E, a_kcontig=1, b_kcontig=0, act=hip.ACT_GELU_ERF)
    if not ok:
        pytest.skip(f"{variant} refuses this problem")
    raw = torch.einsum("erd,edf->erf", X.float(), W1.float())
    _close(Hp, raw, what="fc1 raw")
    _close(H, torch.nn.functional.gelu(raw), what="fc1 gelu")
    O = torch.empty(E, Bk, d, device=dev, dtype=torch.bfloat16)
    assert _run(hip, variant, A=H, B=W2, C=O, M=Bk, N=d, K=f, lda=f, ldb=d, ldc=d, sA=Bk * f, sB=f * d, sC=Bk * d, batch=E,
                a_kcontig=1, b_kcontig=0)
    _close(O, torch.einsum("erf,efd->erd", H.float(), W2.float()), what="fc2")
    dO = torch.randn(E, Bk, d, device=dev).to(torch.bfloat16)
    dHp = torch.empty_like(H)
    ok = _run(hip, variant, A=dO, B=W2, C=dHp, aux=Hp, M=Bk, N=f, K=d, lda=d, ldb=d, ldc=f, ldaux=f, sA=Bk * d, sB=f * d,
              sC=Bk * f, sAux=Bk * f, batch=E, a_kcontig=1, b_kcontig=1, mode=hip.EPI_DACT, act=hip.ACT_GELU_ERF)
    if not ok:      # pp256 / w4 build their operand-carrying epilogues (residual, activation derivative) for interior tiles only
        assert variant in ("pp256", "w4") and (Bk % 256 or f % 256)
        pytest.skip("pp256 refuses the ragged activation-derivative launch (the library's own choice falls back to the 2-stage kernels)")
    xp = Hp.float().requires_grad_(True)
    torch.nn.functional.gelu(xp).sum().backward()
    _close(dHp, torch.einsum("erd,efd->erf", dO.float(), W2.float()) * xp.grad, what="dact")
    dX = torch.empty_like(X)
    assert _run(hip, variant, A=dHp, B=W1, C=dX, M=Bk, N=d, K=f, lda=f, ldb=f, ldc=d, sA=Bk * f, sB=d * f, sC=Bk * d, batch=E,
                a_kcontig=1, b_kcontig=1)
    _close(dX, torch.einsum("erf,edf->erd", dHp.float(), W1.float()), what="fc1 dgrad")
    # md_gemm_args.dact_cached: the forward stores gelu'(pre-activation) in C2, the backward epilogue multiplies by it
    Hd = torch.full_like(H, float("nan"))
    H2 = torch.full_like(H, float("nan"))
    assert _run(hip, variant, A=X, B=W1, C=H2, C2=Hd, M=Bk, N=f, K=d, lda=d, ldb=f, ldc=f, ldc2=f, sA=Bk * d, sB=d * f, sC=Bk * f,
                sC2=Bk * f, batch=E, a_kcontig=1, b_kcontig=0, act=hip.ACT_GELU_ERF, dact_cached=1)
    _close(H2, torch.nn.functional.gelu(raw), what="fc1 gelu (derivative cached)")
    xr = raw.to(torch.bfloat16).float().requires_grad_(True)
    torch.nn.functional.gelu(xr).sum().backward()
    _close(Hd, xr.grad, what="cached gelu'")
    dHc = torch.full_like(H, float("nan"))
    assert _run(hip, variant, A=dO, B=W2, C=dHc, aux=Hd, M=Bk, N=f, K=d, lda=d, ldb=d, ldc=f, ldaux=f, sA=Bk * d, sB=f * d,
                sC=Bk * f, sAux=Bk * f, batch=E, a_kcontig=1, b_kcontig=1, mode=hip.EPI_DACT, act=hip.ACT_GELU_ERF, dact_cached=1)
    _close(dHc, torch.einsum("erd,efd->erf", dO.float(), W2.float()) * Hd.float(), what="dact from the cached derivative")
    _close(dHc, dHp.float(), rel=2e-2, what="cached vs recomputed derivative")
    # dact_cached is refused (-1, nothing launched) with any other activation, or on a forward launch without a C2 to hold the
    # derivative: an inconsistent caller must not get the pre-activation multiplied in silently (every kernel family, one check)
    bad = dict(M=Bk, N=f, K=d, lda=d, ldb=d, ldc=f, ldaux=f, sA=Bk * d, sB=f * d, sC=Bk * f, sAux=Bk * f, batch=E, a_kcontig=1, b_kcontig=1)
    assert hip.gemm(dO, W2, dHc, aux=Hd, mode=hip.EPI_DACT, act=hip.ACT_GELU_TANH, dact_cached=1, expect=None, **bad) == -1
    assert hip.gemm(dO, W2, dHc, act=hip.ACT_GELU_ERF, dact_cached=1, expect=None, **{k: v for k, v in bad.items() if k not in ("ldaux", "sAux")}) == -1


@pytest.mark.parametrize("M,f,d", [(16384, 2816, 1024), (4096, 768, 512), (65536, 2816, 1024)])
def test_swiglu_bwd_fused_in_the_w3_dgrad(hip, M, f, d):
    """MD_EPI_SWIGLU_BWD (ABI 6): da = dy @ W3 with the SwiGLU backward in the epilogue of the 4-wave kernel -- aux = h12 [M, 2f],
    C = dh12 [M, 2f]: dh1 = da * h2 * silu'(h1), dh2 = da * silu(h1) with da = bf16(accumulator) -- against (i) torch fp32 autograd of
    silu(h1) * h2 (dit.py:88-89) and (ii) the two-launch path it replaces (plain data gradient + md_swiglu_bwd), which rounds at the
    same points.  What the kernel does not cover (ragged tiles, a CU hold, any other variant) is refused with NOT_ELIGIBLE and writes
    nothing."""
    torch.manual_seed(M + f)
    L, st = hip.lib(), hip.stream_ptr()
    dy = (torch.randn(M, d, device=dev) * 0.5).to(torch.bfloat16)
    W3 = (torch.randn(d, f, device=dev) * 0.05).to(torch.bfloat16)            # torch layout [out = d, in = f]: the K-strided operand
    h12 = torch.randn(M, 2 * f, device=dev).to(torch.bfloat16)
    dh12 = torch.full((M, 2 * f), float("nan"), device=dev, dtype=torch.bfloat16)
    kw = dict(M=M, N=f, K=d, lda=d, ldb=f, ldc=2 * f, ldaux=2 * f, a_kcontig=1, b_kcontig=0, mode=hip.EPI_SWIGLU_BWD)
    chosen = []
    rc = hip.gemm(dy, W3, dh12, aux=h12, expect=None, chosen=chosen, **kw)
    assert rc == 0 and chosen[-1] == hip.GEMM_W4, (rc, chosen)
    torch.cuda.synchronize()
    # (ii) the two launches
    da = torch.empty(M, f, device=dev, dtype=torch.bfloat16)
    hip.gemm(dy, W3, da, M=M, N=f, K=d, lda=d, ldb=f, ldc=f, a_kcontig=1, b_kcontig=0)
    two = torch.empty_like(dh12)
    hip.check(L.md_swiglu_bwd(da.data_ptr(), f, h12.data_ptr(), 2 * f, two.data_ptr(), 2 * f, M, f, st), "swiglu_bwd")
    torch.cuda.synchronize()
    assert torch.isfinite(dh12.float()).all()
    d2 = (dh12.float() - two.float()).abs()
    assert float((d2 > 0.0079 * two.float().abs() + 1e-6).float().mean()) <= 2e-3, "fused vs two launches: more than 2e-3 of the elements differ by more than a bf16 ulp"
    # (i) torch fp32
    rows = slice(0, 2048)
    hr = h12[rows].float().requires_grad_(True)
    a = torch.nn.functional.silu(hr[:, :f]) * hr[:, f:]
    a.backward(dy[rows].float() @ W3.float())
    _close(dh12[rows], hr.grad, rel=2e-2, what="swiglu bwd fused")
    # refusals: another variant, a CU hold, ragged rows -- nothing is launched, nothing written
    guard = torch.full_like(dh12, 3.0)
    assert hip.gemm(dy, W3, guard, aux=h12, expect=None, variant=hip.GEMM_PP256, **kw) == hip.NOT_ELIGIBLE
    assert hip.gemm(dy, W3, guard, aux=h12, expect=None, cu_limit=248, **kw) == hip.NOT_ELIGIBLE
    kr = dict(kw, M=M - 8)
    assert hip.gemm(dy, W3, guard, aux=h12, expect=None, **kr) == hip.NOT_ELIGIBLE
    torch.cuda.synchronize()
    assert float((guard.float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("Nw,Kw,T,E,ks", [(1024, 1024, 65536, 1, 16), (768, 3072, 65536, 8, 4), (2048, 1024, 78848, 1, 8),
                                          (3072, 1024, 16384, 1, 8), (520, 264, 4096, 3, 4)])
def test_wgrad_splitk(hip, variant, Nw, Kw, T, E, ks):
    """Weight gradients (TN: both operands K-strided, contraction over tokens) as split-K fp32 slices + md_splitk_reduce,
    accumulated into an existing gradient; (1024, 1024, 65536, ks 16) and (768, 3072, 65536, batch 8) are the two
    heaviest weight-gradient launches of the XL/2 step."""
    torch.manual_seed(Nw + Kw + E)
    dY = torch.randn(E, T, Nw, device=dev).to(torch.bfloat16)
    X = torch.randn(E, T, Kw, device=dev).to(torch.bfloat16)
    G = torch.ones(E, Nw, Kw, device=dev)
    ws = torch.full((E * ks * Nw * Kw,), float("nan"), device=dev)
    ok = _run(hip, variant, A=dY, B=X, C=ws, M=Nw, N=Kw, K=T, lda=Nw, ldb=Kw, ldc=Kw, a_kcontig=0, b_kcontig=0,
              mode=hip.EPI_STORE_F32, batch=E, sA=T * Nw, sB=T * Kw, sC=ks * Nw * Kw, sSplit=Nw * Kw, ksplit=ks)
    if not ok:
        pytest.skip(f"{variant} refuses this problem")
    hip.check(hip.lib().md_splitk_reduce(ws.data_ptr(), G.data_ptr(), Nw, Kw, Kw, Nw * Kw, ks, E, 1, hip.stream_ptr()), "reduce")
    torch.cuda.synchronize()
    ref = 1 + torch.einsum("etn,etk->enk", dY.float(), X.float())
    _close(G, ref, rel=2e-3, what=f"{variant} wgrad")


def test_pp256_refuses_what_it_cannot_run(hip):
    A = torch.zeros(256, 192, device=dev, dtype=torch.bfloat16)
    B = torch.zeros(256, 192, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(256, 256, device=dev, dtype=torch.bfloat16)
    # K = 192 is not a multiple of 128; atomics; an activation it has no instantiation for
    assert hip.gemm(A, B, C, 256, 256, 192, lda=192, ldb=192, ldc=256, variant=hip.GEMM_PP256, expect=None) == hip.NOT_ELIGIBLE
    Cf = torch.zeros(256, 256, device=dev)
    assert hip.gemm(A, B, Cf, 256, 256, 128, lda=192, ldb=192, ldc=256, mode=hip.EPI_ATOMIC_F32, ksplit=1,
                    variant=hip.GEMM_PP256, expect=None) == hip.NOT_ELIGIBLE
    assert hip.gemm(A, B, C, 256, 256, 128, lda=192, ldb=192, ldc=256, act=hip.ACT_SILU, variant=hip.GEMM_PP256, expect=None) == hip.NOT_ELIGIBLE
    assert hip.gemm(A, B, C, 256, 256, 128, lda=192, ldb=192, ldc=256, variant=99, expect=None) == -1


@pytest.mark.parametrize("variant,akc,bkc", [("pp256", 1, 1), ("pp256", 1, 0), ("pp256", 0, 0), ("w4", 1, 1), ("w4", 1, 0)])
def test_pp256_race_screen(hip, akc, bkc, variant):
    """The ping-pong kernel orders its LDS ring with counted waits and barriers only: repeat a many-tile launch (several tiles
    per workgroup, short K so tile hand-overs dominate) and require bit-identical results, on an idle chip and while
    another stream streams through HBM (uneven arrival of the DMA pieces)."""
    torch.manual_seed(17)
    M, N, K = 256 * 96 + 40, 1024 + 8, 256
    A, As, lda = _operand(M, K, akc)
    B, Bs, ldb = _operand(N, K, bkc, 0.1)
    f32 = not akc
    outs = []
    noise_stream = torch.cuda.Stream()
    big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
    for rep in range(6):
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        if rep >= 3:
            with torch.cuda.stream(noise_stream):
                for _ in range(4):
                    big.add_(1)
        hip.gemm(As, Bs, C, M, N, K, lda=lda, ldb=ldb, ldc=N, a_kcontig=akc, b_kcontig=bkc,
                 mode=hip.EPI_STORE_F32 if f32 else hip.EPI_STORE_BF16, variant=hip.GEMM_VARIANT_NAMES[variant])
        torch.cuda.synchronize()
        outs.append(C)
    _close(outs[0], A.float() @ B.float().t(), rel=2e-3 if f32 else 2e-2, what=variant)
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), f"{variant} result differs between identical launches (LDS ring race)"


def _time_us(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def test_cu_limit_same_result_and_proportional_time(hip):
    """md_gemm_args.cu_limit (data parallelism: the CUs of RCCL's channels are left out of the persistent kernel's grid while a
    collective is in flight): the same plan on 256 - k workgroups gives bit-identical results, and costs what the smaller chip
    costs -- time <= 256 / (256 - k) x 1.1 of the full-grid launch on a launch with many tiles per workgroup (12), where
    whole-tile granularity does not dominate (1024 tiles on 248 workgroups are 5 rounds instead of 4 whatever the scheduler)."""
    M, N, K, k = 65536, 3072, 1024, 8
    torch.manual_seed(7)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    C0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C1 = torch.empty_like(C0)
    kw = dict(A=A, B=B, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, variant=hip.GEMM_PP256)
    t_full = _time_us(lambda: hip.gemm(C=C0, **kw))
    t_lim = _time_us(lambda: hip.gemm(C=C1, cu_limit=256 - k, **kw))
    assert torch.equal(C0, C1)
    _close(C1[:512], A[:512].float() @ B.float().t(), what="cu_limit 248")
    bound = 256 / (256 - k) * 1.1
    print(f"cu_limit: full grid {t_full:.1f} us, {256 - k} workgroups {t_lim:.1f} us (x{t_lim / t_full:.3f}, bound x{bound:.3f})")
    assert t_lim <= bound * t_full, (t_lim, t_full)


def test_cu_limit_under_a_resident_collective(hip):
    """What cu_limit is FOR, emulated on one GPU: 8 workgroups of a spinning kernel (tests/probes: 96 KiB of LDS each, so a
    persistent-GEMM workgroup cannot share their CU -- an RCCL kernel with 8 channels) are resident on a side stream while
    the GEMM runs.  With the full grid the 8 GEMM workgroups that find their CU taken start only when the hog leaves or another
    workgroup finishes its whole tile list; with cu_limit = 248 every tile is dealt to a free CU.  Reported, and asserted only in
    the direction that matters: the limited grid must not be slower than the full one while the CUs are held."""
    from tests import probes
    M, N, K = 65536, 1024, 1024                     # 4 tiles per workgroup: the full grid's late workgroups cost a whole extra pass
    torch.manual_seed(8)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(A=A, B=B, C=C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, variant=hip.GEMM_PP256)
    scratch = torch.zeros(1, device=dev, dtype=torch.int32)
    side = torch.cuda.Stream()
    t_free = _time_us(lambda: hip.gemm(**kw), reps=4)
    res = {}
    for name, lim in (("full", 0), ("limited", 248)):
        times = []
        for _ in range(7):
            torch.cuda.synchronize()
            with torch.cuda.stream(side):            # the hog is resident for 2 ms: far longer than one GEMM launch
                hip.check(probes.lib().mdp_cu_hog(8, 2000, scratch.data_ptr(), side.cuda_stream), "hog")
            torch.cuda._sleep(200000)                # let the hog's workgroups take their CUs first
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.gemm(cu_limit=lim, **kw)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        # the MEDIAN: a trial in which the hog was not resident yet measures the free chip, and the minimum would pick exactly
        # those trials (round 5: "full" 121 us = the free-chip time, against 195 us with the CUs really held)
        res[name] = sorted(times)[len(times) // 2]
    print(f"cu_limit under 8 held CUs: free chip {t_free:.1f} us | full grid {res['full']:.1f} us | 248 workgroups {res['limited']:.1f} us")
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/cu_limit_under_hog.json", "w") as fh:
        json.dump({"shape": [M, N, K], "held_cus": 8, "free_chip_us": t_free, "full_grid_us": res["full"], "limited_248_us": res["limited"]}, fh)
    if res["full"] < 1.1 * t_free:
        pytest.skip(f"the hog kernel did not hold its CUs while the GEMM ran (full grid {res['full']:.1f} us ~ free chip {t_free:.1f} us): nothing to compare")
    assert res["limited"] <= 1.05 * res["full"], res
