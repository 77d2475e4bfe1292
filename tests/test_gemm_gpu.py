"""GEMM parity (HIP MFMA kernel vs torch fp32 matmul of the same bf16-rounded operands) — all four operand
layouts, ragged sizes, every epilogue.  Tolerance: fp32-accumulated bf16 products, outputs rounded to bf16:
|err| <= 2e-2 * max|ref| (bf16 has 8 mantissa bits; K up to 1024)."""
import pytest
import torch

from tests import probes

pytestmark = pytest.mark.gpu


def _mk(rows, k, kcontig, dev, scale=1.0):
    t = (torch.randn(rows, k, device=dev) * scale).to(torch.bfloat16)
    if kcontig:
        return t, t.contiguous(), k            # stored [rows, k], ld = k
    tt = t.t().contiguous()                    # stored [k, rows], ld = rows
    return t, tt, rows


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (384, 128, 64), (200, 72, 104), (1024, 768, 1024), (128, 8, 512)])
def test_gemm_layouts(hip, akc, bkc, M, N, K):
    dev = "cuda"
    torch.manual_seed(M * 7 + N * 3 + K + akc * 2 + bkc)
    A, As, lda = _mk(M, K, akc, dev)
    B, Bs, ldb = _mk(N, K, bkc, dev)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    hip.gemm(As, Bs, C, M, N, K, lda=lda, ldb=ldb, ldc=N, a_kcontig=akc, b_kcontig=bkc)
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t()
    err = (C.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, f"max err {err} vs max {ref.abs().max().item()}"


def test_mfma_probe(hip):
    dev = "cuda"
    A = torch.randn(32, 16, device=dev).to(torch.bfloat16)
    B = torch.randn(16, 32, device=dev).to(torch.bfloat16)
    D = torch.zeros(32, 32, device=dev)
    hip.check(probes.lib().mdp_mfma_probe(A.data_ptr(), B.data_ptr(), D.data_ptr(), hip.stream_ptr()), "probe")
    torch.cuda.synchronize()
    ref = A.float() @ B.float()
    assert (D - ref).abs().max().item() < 1e-3


def test_tr_read_semantics(hip):
    """ds_read_b64_tr_b16: per 16-lane group, lane i supplies the address of 4 contiguous elements forming
    row (i / 4), cols 4 * (i % 4).. of a [4][16] block; lane i receives column i (4 rows)."""
    dev = "cuda"
    pitch = 160
    addr = torch.zeros(64, dtype=torch.int32)
    for l in range(64):
        li, g = l & 15, l >> 4
        addr[l] = (li >> 2) * pitch + g * 16 + (li & 3) * 4
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    a = addr.to(dev)
    hip.check(probes.lib().mdp_tr_probe(a.data_ptr(), out.data_ptr(), hip.stream_ptr()), "tr probe")
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4)
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        li, g = l & 15, l >> 4
        for j in range(4):
            exp[l, j] = j * pitch + g * 16 + li
    print("tr-read got:\n", got[:20].tolist())
    assert torch.equal(got, exp), f"tr-read semantics differ; lanes 0..19 got {got[:20].tolist()}"


def test_gemm_epilogues(hip):
    dev = "cuda"
    torch.manual_seed(3)
    M, N, K = 512, 256, 320
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ref = A.float() @ B.float().t() + bias
    # bias + gelu_tanh, with raw copy
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty_like(C)
    hip.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, act=hip.ACT_GELU_TANH, C2=C2, ldc2=N)
    torch.cuda.synchronize()
    r2 = torch.nn.functional.gelu(ref, approximate="tanh")
    assert (C.float() - r2).abs().max() < 3e-2
    assert (C2.float() - ref).abs().max() < 3e-2
    # gated residual
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    rows_per_sample = 64
    gate = torch.randn(M // rows_per_sample, N, device=dev).to(torch.bfloat16)
    out = torch.empty_like(res)
    hip.gemm(A, B, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, mode=hip.EPI_RESIDUAL, res=res, ldr=N,
             gate=gate, ldg=N, rows_per_sample=rows_per_sample)
    torch.cuda.synchronize()
    r3 = res.float() + gate.float().repeat_interleave(rows_per_sample, 0) * ref
    assert (out.float() - r3).abs().max() < 6e-2
    # fp32 store / accumulate / split-K atomics
    Cf = torch.zeros(M, N, device=dev)
    hip.gemm(A, B, Cf, M, N, K, lda=K, ldb=K, ldc=N, mode=hip.EPI_STORE_F32)
    hip.gemm(A, B, Cf, M, N, K, lda=K, ldb=K, ldc=N, mode=hip.EPI_ACCUM_F32)
    hip.gemm(A, B, Cf, M, N, K, lda=K, ldb=K, ldc=N, mode=hip.EPI_ATOMIC_F32, ksplit=3)
    torch.cuda.synchronize()
    r4 = 3 * (A.float() @ B.float().t())
    assert (Cf - r4).abs().max() < 1e-2
    # dact
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    Cd = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    hip.gemm(A, B, Cd, M, N, K, lda=K, ldb=K, ldc=N, mode=hip.EPI_DACT, act=hip.ACT_GELU_ERF, aux=aux, ldaux=N)
    torch.cuda.synchronize()
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    r5 = (A.float() @ B.float().t()) * x.grad
    assert (Cd.float() - r5).abs().max() < 3e-2


def test_gemm_batched_grouped(hip):
    """8 'experts': per-expert A [rows, d] and K-strided weights [d, f] (MoE layout, dit.py:121-122)."""
    dev = "cuda"
    torch.manual_seed(5)
    E, R, D, F = 8, 192, 128, 256
    X = torch.randn(E, R, D, device=dev).to(torch.bfloat16)
    W = (torch.randn(E, D, F, device=dev) * 0.1).to(torch.bfloat16)
    H = torch.empty(E, R, F, device=dev, dtype=torch.bfloat16)
    hip.gemm(X, W, H, R, F, D, lda=D, ldb=F, ldc=F, a_kcontig=1, b_kcontig=0, batch=E, sA=R * D, sB=D * F,
             sC=R * F)
    torch.cuda.synchronize()
    ref = torch.einsum("erd,edf->erf", X.float(), W.float())
    assert (H.float() - ref).abs().max() < 2e-2 * ref.abs().max()


def test_gemm_splitk_workspace_reduce(hip):
    """Split-K through fp32 workspace slices + md_splitk_reduce (the weight-gradient path): TN layout, batched."""
    dev = "cuda"
    torch.manual_seed(9)
    E, R, D, F = 3, 2048, 256, 384
    X = torch.randn(E, R, D, device=dev).to(torch.bfloat16)       # [rows, d]
    dH = torch.randn(E, R, F, device=dev).to(torch.bfloat16)      # [rows, f]
    G = torch.ones(E, D, F, device=dev)                           # accumulate into existing grads
    ks = 4
    ws = torch.empty(E * ks * D * F, device=dev)
    hip.gemm(X, dH, ws, D, F, R, lda=D, ldb=F, ldc=F, a_kcontig=0, b_kcontig=0, mode=hip.EPI_STORE_F32, batch=E,
             sA=R * D, sB=R * F, sC=ks * D * F, sSplit=D * F, ksplit=ks)
    hip.check(hip.lib().md_splitk_reduce(ws.data_ptr(), G.data_ptr(), D, F, F, D * F, ks, E, 1, hip.stream_ptr()), "reduce")
    torch.cuda.synchronize()
    ref = 1 + torch.einsum("erd,erf->edf", X.float(), dH.float())
    assert (G - ref).abs().max() < 2e-3 * ref.abs().max()


@pytest.mark.parametrize("akc,bkc", [(1, 1), (1, 0)])
def test_gemm_many_tiles_short_k(hip, akc, bkc):
    """Large launches (thousands of tiles, K ~ 1024) as the XL/2 microbatches produce them: ragged M / N / K edges, the
    gated-residual epilogue, and a grouped (batched) launch with GELU + raw copy, against torch fp32 matmul."""
    dev = "cuda"
    torch.manual_seed(11 + akc + 2 * bkc)
    M, N, K = 32768 + 72, 1024 + 40, 1024 + 24          # 129 x 5 = 645 tiles of 256^2, ragged in every dimension
    A, As, lda = _mk(M, K, akc, dev)
    B, Bs, ldb = _mk(N, K, bkc, dev, scale=0.05)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    rps = 8
    gate = torch.randn(M // rps, N, device=dev).to(torch.bfloat16)
    out = torch.empty_like(res)
    hip.gemm(As, Bs, out, M, N, K, lda=lda, ldb=ldb, ldc=N, a_kcontig=akc, b_kcontig=bkc, mode=hip.EPI_RESIDUAL, res=res,
             ldr=N, gate=gate, ldg=N, rows_per_sample=rps)
    torch.cuda.synchronize()
    ref = res.float() + gate.float().repeat_interleave(rps, 0) * (A.float() @ B.float().t()).to(torch.bfloat16).float()
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item() + 1e-3, err
    # grouped: 8 problems x 68 tiles, K = 1024, plain bf16 store with GELU and the raw copy
    E, R, D, F = 8, 4224, 1024, 1024
    X = torch.randn(E, R, D, device=dev).to(torch.bfloat16)
    W = (torch.randn(E, F, D, device=dev) * 0.03).to(torch.bfloat16)
    H = torch.empty(E, R, F, device=dev, dtype=torch.bfloat16)
    H2 = torch.empty_like(H)
    hip.gemm(X, W, H, R, F, D, lda=D, ldb=D, ldc=F, a_kcontig=1, b_kcontig=1, batch=E, sA=R * D, sB=F * D, sC=R * F,
             act=hip.ACT_GELU_ERF, C2=H2, ldc2=F, sC2=R * F)
    torch.cuda.synchronize()
    raw = torch.einsum("erd,efd->erf", X.float(), W.float())
    assert (H2.float() - raw).abs().max() <= 2e-2 * raw.abs().max()
    assert (H.float() - torch.nn.functional.gelu(raw)).abs().max() <= 2e-2 * raw.abs().max()


def test_vmcnt_retires_loads_and_stores_in_issue_order(hip):
    """The persistent GEMM counts its vector-memory operations by hand (s_waitcnt vmcnt(N) against the DMA ring and the prefetched
    epilogue operands).  Its counts are exact only if a wave's loads and stores retire the counter in issue order — what
    LLVM's own waitcnt insertion assumes on gfx9.  Probe: per lane one cold load (its own cache line of a 1 GiB buffer), four
    hot stores, s_waitcnt vmcnt(4); a lane whose load had not landed would prove out-of-order retirement."""
    dev = "cuda"
    cold = torch.empty(1 << 28, device=dev, dtype=torch.float32).normal_()         # 1 GiB, far larger than L2 + Infinity Cache
    hot = torch.zeros(4096, device=dev, dtype=torch.float32)
    stale = torch.zeros(1, device=dev, dtype=torch.int32)
    flush = torch.empty(1 << 27, device=dev, dtype=torch.float32)
    for rep in range(8):
        flush.normal_()                                                              # evict `cold` from the caches between rounds
        hip.check(probes.lib().mdp_vmcnt_order_probe(cold.data_ptr(), cold.numel() * 4, hot.data_ptr(), stale.data_ptr(), 4096,
                                                       hip.stream_ptr()), "probe")
    torch.cuda.synchronize()
    assert int(stale.item()) == 0, f"{int(stale.item())} lanes saw a store retire ahead of an older load"
