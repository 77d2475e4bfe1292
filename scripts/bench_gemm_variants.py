"""Per-shape TFLOP/s of every md_gemm_bf16 kernel (forced through md_gemm_args.variant) on the heaviest GEMM launches of
one MicroDiT-XL/2 microbatch (shape list = profiles/r1_gemm_final_shape_tables.txt, microbatch 1024 and 256).  HIP events
on the launch stream, random operands, interleaved rounds (guide rule 24).  Usage: python scripts/bench_gemm_variants.py [--mb 1024]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from micro_diffusion_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--variants", default="auto,reg128,dma128,paced256,pp256")
args = ap.parse_args()
S = args.mb * 64          # backbone tokens
C = args.mb * 77          # caption tokens
T = args.mb * 256         # mixer tokens
dev = "cuda"
# (M, N, K, batch, akc, bkc, ksplit, mode, label)   mode: b = bf16 store, r = gated residual, g = gelu + C2, d = dact, f = f32 slices
shapes = [
    (S, 1024, 1024, 1, 1, 1, 1, "b", "bb proj/q fwd"),
    (S, 1024, 1024, 1, 1, 1, 1, "r", "bb proj fwd (gated res)"),
    (S, 1024, 1024, 1, 1, 0, 1, "b", "bb dgrad 1024"),
    (C, 2048, 1024, 1, 1, 1, 1, "b", "kv_linear fwd"),
    (C, 1024, 2048, 1, 1, 0, 1, "b", "kv dgrad"),
    (S, 3072, 1024, 1, 1, 1, 1, "b", "bb qkv fwd"),
    (S, 1024, 3072, 1, 1, 0, 1, "b", "bb qkv dgrad"),
    (S, 5632, 1024, 1, 1, 1, 1, "b", "ffn w12 fwd"),
    (S, 1024, 2816, 1, 1, 1, 1, "r", "ffn w3 fwd (gated res)"),
    (S // 4, 3840, 1024, 8, 1, 0, 1, "g", "moe fc1 fwd f=3840"),
    (S // 4, 1024, 3840, 8, 1, 0, 1, "b", "moe fc2 fwd f=3840"),
    (S // 4, 3840, 1024, 8, 1, 1, 1, "d", "moe fc2 dgrad (dact)"),
    (T // 4, 3072, 768, 8, 1, 0, 1, "g", "mixer moe fc1 fwd"),
    (T // 4, 768, 3072, 8, 1, 0, 1, "b", "mixer moe fc2 fwd"),
    (T, 768, 768, 1, 1, 1, 1, "b", "mixer proj fwd"),
    (T, 2304, 768, 1, 1, 1, 1, "b", "mixer qkv fwd"),
    (T, 768, 2304, 1, 1, 0, 1, "b", "mixer qkv dgrad"),
    (1024, 1024, S, 1, 0, 0, 16, "f", "wgrad 1024x1024 ks16"),
    (2048, 1024, C, 1, 0, 0, 8, "f", "wgrad kv ks8"),
    (768, 3072, T // 4, 8, 0, 0, 6 if args.mb == 1024 else 3, "f", "wgrad mixer moe"),
    (3072, 1024, S, 1, 0, 0, 5, "f", "wgrad qkv ks5"),
    (768, 768, T, 1, 0, 0, 28, "f", "wgrad mixer 768 ks28"),
    (8192, 8192, 8192, 1, 1, 1, 1, "b", "8k cube"),
    (4096, 4096, 4096, 1, 1, 1, 1, "b", "4k cube"),
]
names = args.variants.split(",")
print(f"# microbatch {args.mb}; TFLOP/s median of {args.rounds} interleaved rounds x {args.iters} launches; '-' = variant refuses the problem")
print(f"{'shape':28s} {'M':>7} {'N':>5} {'K':>7} {'bat':>3} {'AB':>2} {'ks':>3} {'mode':>4} " + " ".join(f"{n:>9}" for n in names))
for M, N, K, bt, akc, bkc, ks, mode, label in shapes:
    if ks > 1 and (K // ks) % 128:
        ks_eff = max(1, K // (128 * max(1, (K // ks) // 128)))
        while K % ks_eff or (K // ks_eff) % 128:
            ks_eff -= 1
        ks = ks_eff
    A = torch.randn((bt, M, K) if akc else (bt, K, M), device=dev).to(torch.bfloat16)
    B = (torch.randn((bt, N, K) if bkc else (bt, K, N), device=dev) * 0.05).to(torch.bfloat16)
    f32 = mode == "f"
    Cc = torch.empty(bt * ks, M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(lda=K if akc else M, ldb=K if bkc else N, ldc=N, a_kcontig=akc, b_kcontig=bkc, batch=bt, sA=M * K, sB=N * K,
              sC=ks * M * N, ksplit=ks)
    if mode == "r":
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        gate = torch.randn(M // 64, N, device=dev).to(torch.bfloat16)
        c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw.update(mode=hip.EPI_RESIDUAL, res=res, ldr=N, gate=gate, ldg=N, rows_per_sample=64, C2=c2, ldc2=N)
    elif mode == "g":
        c2 = torch.empty(bt, M, N, device=dev, dtype=torch.bfloat16)
        kw.update(act=hip.ACT_GELU_ERF, C2=c2, ldc2=N, sC2=M * N)
    elif mode == "d":
        aux = torch.randn(bt, M, N, device=dev).to(torch.bfloat16)
        kw.update(mode=hip.EPI_DACT, act=hip.ACT_GELU_ERF, aux=aux, ldaux=N, sAux=M * N)
    elif mode == "f":
        kw.update(mode=hip.EPI_STORE_F32, sSplit=M * N)
    times = {n: [] for n in names}
    ok = {}
    for n in names:
        ok[n] = hip.gemm(A, B, Cc, M, N, K, variant=hip.GEMM_VARIANT_NAMES[n], expect=None, **kw) == 0
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for n in names:
            if not ok[n]:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                hip.gemm(A, B, Cc, M, N, K, variant=hip.GEMM_VARIANT_NAMES[n], **kw)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / args.iters)
    fl = 2.0 * M * N * K * bt
    row = []
    for n in names:
        if not ok[n]:
            row.append(f"{'-':>9}")
        else:
            t = sorted(times[n])[len(times[n]) // 2]
            row.append(f"{fl / t / 1e9:9.0f}")
    print(f"{label:28s} {M:7d} {N:5d} {K:7d} {bt:3d} {akc}{bkc:1d} {ks:3d} {mode:>4} " + " ".join(row), flush=True)
    del A, B, Cc
