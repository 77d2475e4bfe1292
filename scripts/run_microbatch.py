"""A few XL/2 microbatches (fwd+bwd) for rocprofv3 runs: python scripts/run_microbatch.py [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd.model import create_latent_diffusion
import bench
torch.manual_seed(18)
model = create_latent_diffusion(dit_arch="MicroDiT_XL_2", latent_res=32, train_mask_ratio=0.75)
model.dit.to("cuda"); bench.dezero_(model.dit); model.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator(device="cuda").manual_seed(1)
batch = {"image_latents": (torch.randn(B, 4, 32, 32, device="cuda", generator=g) * 0.8).half(),
         "caption_latents": torch.randn(B, 1, 77, 1024, device="cuda", generator=g).half(),
         "drop_caption_mask": torch.ones(B, device="cuda")}
model(batch)[0].backward(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    model(batch)[0].backward()
e1.record(); torch.cuda.synchronize()
print(f"microbatch {B}: {e0.elapsed_time(e1)/iters:.2f} ms per fwd+bwd")
