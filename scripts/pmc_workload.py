"""Workload for the rocprofv3 --pmc passes behind bench.py's roofline.traffic: a calibration copy of known size (torch
elementwise copy, 16-byte accesses: 1 GiB read + 1 GiB written per launch) followed by one forward + backward of a
MicroDiT-XL/2 microbatch (the GEMM launches of the headline step).  Run once per counter:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python scripts/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python scripts/pmc_workload.py
then scripts/pmc_traffic.py turns the two CSVs into profiles/r4_gemm_traffic.json (stamped with the library's source hash)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from micro_diffusion_amd.model import create_latent_diffusion  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = torch.empty(1 << 28, device="cuda", dtype=torch.float32).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)                     # calibration launches: exactly 2^30 bytes read, 2^30 written each
torch.cuda.synchronize()
del src, dst
torch.manual_seed(18)
model = create_latent_diffusion(dit_arch="MicroDiT_XL_2", latent_res=32, train_mask_ratio=0.75)
model.dit.to("cuda")
bench.dezero_(model.dit)
model.train()
g = torch.Generator(device="cuda").manual_seed(1)
batch = {"image_latents": (torch.randn(mb, 4, 32, 32, device="cuda", generator=g) * 0.8).half(),
         "caption_latents": torch.randn(mb, 1, 77, 1024, device="cuda", generator=g).half(),
         "drop_caption_mask": torch.ones(mb, device="cuda")}
model.train_microbatch(batch)           # the Trainer's autograd-free microbatch: the launch sequence bench.py times
torch.cuda.synchronize()
print("pmc workload done")
