"""Per-tensor difference of the staged bf16 gradient: one-microbatch step with / without DiTEngine.wgrad_bf16 (one rank over RCCL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_dp_gpu as T
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from micro_diffusion_amd.trainer import Trainer
res = {}
for store in (False, True):
    model, opt, tr, part = T._build((0, T.BATCH), "bf16", T.BATCH)
    tr = Trainer(model, opt, tr.schedule, clip_norm=0.25, microbatch_size=T.BATCH, exchange="bf16", single_rank_exchange=True, dp_mode="sharded")
    tr.sync.store_bf16 = store
    tr.train_step(part)
    torch.cuda.synchronize()
    f = model.dit.flat_buffers()
    res[store] = {n: tr.sync.gbf[o:o + f["P"][n].numel()].float().cpu() for n, o in f["offs"].items() if n in f["P"]}
    print("store", store, "stored tensors", tr.sync.last_stored, "gnorm", float(opt.grad_norm().item()))
for n in res[False]:
    a, b = res[False][n], res[True][n]
    e = float((a - b).norm() / (a.norm() + 1e-30))
    if e > 1e-2:
        print(f"{n:60s} shape {tuple(model.dit.flat_buffers()['P'][n].shape)}  rel {e:.3e}  |a| {float(a.norm()):.3e} |b| {float(b.norm()):.3e}")
dist.destroy_process_group()
