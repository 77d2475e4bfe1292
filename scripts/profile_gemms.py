"""Per-shape breakdown of the GEMM launches of one XL/2 microbatch (fwd+bwd), from per-launch HIP events."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd.model import create_latent_diffusion
import bench

torch.manual_seed(18)
model = create_latent_diffusion(dit_arch="MicroDiT_XL_2", latent_res=32, train_mask_ratio=0.75)
model.dit.to("cuda"); bench.dezero_(model.dit); model.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(1)
batch = {"image_latents": (torch.randn(B, 4, 32, 32, device="cuda", generator=g) * 0.8).half(),
         "caption_latents": torch.randn(B, 1, 77, 1024, device="cuda", generator=g).half(),
         "drop_caption_mask": torch.ones(B, device="cuda")}
for _ in range(2):
    model.train_microbatch(batch)
torch.cuda.synchronize()
eng = model.dit.engine
eng.gemm_profile = []
eng.kernel_profile = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); model.train_microbatch(batch); e1.record(); torch.cuda.synchronize()
prof, eng.gemm_profile = eng.gemm_profile, None
kprof, eng.kernel_profile = eng.kernel_profile, None
for name, recs in kprof.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in recs)
    print(f"{name:16s} {len(recs):5d} launches {ms:8.2f} ms  {sum(r[2] for r in recs) / ms / 1e9:7.2f} TB/s (algorithmic)")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for a, b, fl, key, _ in prof:
    r = agg[key]; r[0] += 1; r[1] += a.elapsed_time(b); r[2] += fl
tot = sum(r[1] for r in agg.values())
print(f"microbatch {B}: total (with event overhead) {e0.elapsed_time(e1):.1f} ms, gemm {tot:.1f} ms, launches {len(prof)}")
print(f"{'M':>7} {'N':>6} {'K':>7} {'bat':>3} {'A':>1}{'B':>1} {'ks':>3} {'cnt':>4} {'ms':>8} {'%':>5} {'TF/s':>7}")
for key, r in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    M, N, K, bt, ak, bk, ks = key
    print(f"{M:7d} {N:6d} {K:7d} {bt:3d} {ak}{bk} {ks:3d} {r[0]:4d} {r[1]:8.2f} {100*r[1]/tot:5.1f} {r[2]/r[1]/1e9:7.1f}")
kinds = collections.defaultdict(lambda: [0.0, 0.0])
for key, r in agg.items():
    M, N, K, bt, ak, bk, ks = key
    kind = ("wgrad" if (ak == 0 and bk == 0) else "dgrad/NN" if (ak == 1 and bk == 0) else "fwd/NT" if (ak == 1 and bk == 1) else "TN") + (" grouped" if bt > 1 else "")
    kinds[kind][0] += r[1]; kinds[kind][1] += r[2]
for k, (ms, fl) in kinds.items():
    print(f"{k:18s} {ms:8.2f} ms  {fl/ms/1e9:7.1f} TF/s")
