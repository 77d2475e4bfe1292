"""Functional smoke of the four stage configs at full XL/2 width on the GPU (tiny batches): one optimisation step each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_diffusion_amd import config as mdcfg
from micro_diffusion_amd.trainer import FusedAdamW, Trainer
import bench
for name in ("res_256_pretrain", "res_256_finetune", "res_512_pretrain", "res_512_finetune"):
    cfg = mdcfg.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs"), name + ".yaml")
    torch.manual_seed(cfg["seed"])
    model = mdcfg.instantiate(cfg["model"]); model.dit.to("cuda"); bench.dezero_(model.dit); model.train()
    tr = Trainer(model, FusedAdamW(model.dit, lr=cfg["optimizer"]["lr"]), None, clip_norm=cfg["algorithms"]["gradient_clipping"]["clip_norm"],
                 microbatch_size=min(16, cfg["trainer"]["device_train_microbatch_size"]))
    B, res = 32, cfg["model"]["latent_res"]
    g = torch.Generator(device="cuda").manual_seed(3)
    batch = {"image_latents": (torch.randn(B, 4, res, res, device="cuda", generator=g) * 0.8).half(),
             "caption_latents": torch.randn(B, 1, 77, 1024, device="cuda", generator=g).half(),
             "drop_caption_mask": (torch.rand(B, device="cuda", generator=g) >= 0.1).float()}
    losses = []
    for _ in range(2):
        batch["caption_latents"] = torch.randn(B, 1, 77, 1024, device="cuda", generator=g).half()
        losses.append(float(tr.train_step(batch)))
    torch.cuda.synchronize()
    assert all(l == l and l < 1e4 for l in losses), losses
    print(f"{name}: mask {cfg['model']['train_mask_ratio']} res {res} losses {losses} peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    del model, tr
    torch.cuda.empty_cache()
