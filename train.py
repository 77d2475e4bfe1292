"""Training entry point with the reference's CLI (reference train.py:14-123, README.md:34-50):

    python train.py --config-path ./configs --config-name res_256_pretrain.yaml exp_name=... model.train_mask_ratio=0.75
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config-path ... (one rank / GPU)

Composer / hydra are replaced by micro_diffusion_amd.{config,trainer}; the model, loss, backward, gradient averaging,
clipping and AdamW all run as HIP kernels."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from micro_diffusion_amd import config as mdcfg  # noqa: E402
from micro_diffusion_amd.model import text_encoder_embedding_format  # noqa: E402
from micro_diffusion_amd.trainer import FusedAdamW, LRSchedule, Trainer, parse_batches  # noqa: E402


def train(cfg: dict):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from micro_diffusion_amd.trainer import cap_rccl_channels
        cap_rccl_channels()       # NCCL_MAX_NCHANNELS before RCCL starts: the CUs its kernels hold are left out of the GEMM grids
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.manual_seed(cfg["seed"])                               # reproducibility.seed_all(cfg.seed)  (train.py:23)
    assert cfg["model"]["precomputed_latents"], "latents must be precomputed (train.py:25)"
    model = mdcfg.instantiate(cfg["model"])
    model.dit.to("cuda")
    model.train()
    # Same init on every rank (built under the shared seed, there is no weight broadcast), then rank-wise noise: Composer's
    # Trainer re-seeds each rank with seed + rank after the model exists, so the ranks draw different
    # sigma / eps / mask noise for their shards of the global batch.
    torch.manual_seed(cfg["seed"] + rank)
    carried_opt = None
    if cfg["trainer"].get("load_path"):
        ckpt = torch.load(cfg["trainer"]["load_path"], map_location="cuda")
        sd = ckpt.get("state", {}).get("model", ckpt)
        # Stage hand-off (configs/res_256_finetune.yaml:92-97): without `load_weights_only` Composer also restores the AdamW
        # moments; the ignored keys are the stored learning rates, i.e. the new stage's lr / schedule come from ITS config.
        if isinstance(ckpt.get("optimizer"), dict) and not cfg["trainer"].get("load_weights_only", False):
            carried_opt = ckpt["optimizer"]
        sd = {k[len("dit."):] if k.startswith("dit.") else k: v for k, v in sd.items()}
        ignore = [k.split("/")[-1] for k in cfg["trainer"].get("load_ignore_keys", [])]
        sd = {k: v for k, v in sd.items() if not any(k == i.replace("dit.", "") for i in ignore)}
        model.dit.load_state_dict(sd, strict=bool(cfg["trainer"].get("load_strict_model_weights", True)) and not ignore)
    ocfg = dict(cfg["optimizer"])
    ocfg.pop("_target_")
    # EMA of the weights (configs/res_512_*.yaml:4-9, diffusion.algorithms.ema.EMA): folded into the AdamW kernel.  The
    # reference's own train.py never instantiates it ("Algorithm ema not supported", train.py:87-88); here it is honoured.
    ema_cfg = (cfg.get("algorithms") or {}).get("ema")
    ema_kw = {}
    if ema_cfg:
        if ema_cfg.get("half_life") is not None or parse_batches(ema_cfg.get("update_interval", "1ba")) != 1:
            raise ValueError("ema: only `smoothing` with update_interval 1ba is supported (the values the stage configs use)")
        ema_kw = dict(ema_smoothing=float(ema_cfg["smoothing"]), ema_start=parse_batches(ema_cfg.get("ema_start", "0ba")))
    opt = FusedAdamW(model.dit, lr=ocfg["lr"], betas=tuple(ocfg.get("betas", (0.9, 0.999))), eps=ocfg.get("eps", 1e-8),
                     weight_decay=ocfg.get("weight_decay", 0.0), **ema_kw)
    if carried_opt is not None:
        opt.load_state_dict(carried_opt)          # keyed by parameter name; raises on a mismatch with this model
    max_ba = parse_batches(cfg["trainer"]["max_duration"])
    scfg = dict(cfg["scheduler"])
    sched = LRSchedule.from_target(scfg.pop("_target_"), t_max=max_ba, **scfg)
    clip = 0.0
    for name, alg in (cfg.get("algorithms") or {}).items():
        if name == "gradient_clipping":
            clip = float(alg["clip_norm"])
        elif name == "ema":
            pass                                                   # handled above (fused into the optimiser kernel)
        elif name != "low_precision_layernorm":                    # LP-LayerNorm is the engine's native numerics
            print(f"Algorithm {name} not supported.")              # same message as the reference (train.py:87-88)
    seq, emb = text_encoder_embedding_format(cfg["model"]["text_encoder_name"])
    ds = cfg["dataset"]
    loader = mdcfg.instantiate(ds["train"], image_size=ds["image_size"], batch_size=ds["train_batch_size"] // world,
                               cap_seq_size=seq, cap_emb_dim=emb, cap_drop_prob=ds["cap_drop_prob"])
    trainer = Trainer(model, opt, sched, clip_norm=clip, microbatch_size=int(cfg["trainer"]["device_train_microbatch_size"]))
    save_every = parse_batches(cfg["trainer"].get("save_interval", "0ba"))
    folder = cfg["trainer"].get("save_folder")
    log_every = int(cfg.get("misc", {}).get("log_interval", 10))
    start = 0
    latest = os.path.join(folder, "latest.pt") if folder else None
    if cfg["trainer"].get("autoresume") and latest and os.path.exists(latest):
        # Composer's autoresume: continue the run whose checkpoints live in save_folder (weights, AdamW moments, batch
        # counter = LR-schedule position, and the data position: the loader's order is a function of (seed, epoch)).
        ck = torch.load(latest, map_location="cuda")
        model.dit.load_state_dict({k[len("dit."):]: v for k, v in ck["state"]["model"].items()})
        opt.load_state_dict(ck["optimizer"])
        start = int(ck["batch"])
        trainer.batches_seen = start
        if ck.get("loader") is not None and hasattr(loader, "load_state_dict"):
            loader.load_state_dict(ck["loader"])
        # the noise stream must not restart from the initial seed: fold the position into it (every rank can do this
        # without having saved its own generator state; rank 0's exact state is restored when present)
        torch.manual_seed(cfg["seed"] + rank + 1000003 * start)
        if rank == 0 and ck.get("rng_cuda") is not None:
            torch.cuda.set_rng_state(ck["rng_cuda"].cpu())
        if rank == 0:
            print(json.dumps({"resumed_from": latest, "batch": start}), flush=True)
    eval_every = parse_batches(cfg["trainer"].get("eval_interval", "0ba") or "0ba")
    eval_loader = None
    if eval_every and ds.get("eval"):
        eval_loader = mdcfg.instantiate(ds["eval"], image_size=ds["image_size"], batch_size=ds["eval_batch_size"] // world,
                                        cap_seq_size=seq, cap_emb_dim=emb, loop=False)
    trainer.sync_replicas()        # rank 0's weights / moments everywhere (same-seed init and resume make them equal already)
    check_every = int(cfg.get("misc", {}).get("replica_check_interval", 500))
    t_last = time.time()
    for step, batch in zip(range(start, max_ba), loader):
        loss = trainer.train_step(batch)
        if eval_loader is not None and (step + 1) % eval_every == 0:
            trainer.consolidate()                                  # sharded optimiser: whole fp32 weights / EMA on every rank
            ev = evaluate(model, eval_loader, world, microbatch=trainer.microbatch_size, opt=opt)
            if rank == 0:
                print(json.dumps({"batch": step + 1, "metrics/eval/loss": ev}), flush=True)
        if world > 1 and check_every and (step + 1) % check_every == 0 and not trainer.replicas_in_sync():
            raise RuntimeError(f"data-parallel replicas diverged at batch {step + 1} (weight checksums differ across ranks)")
        if not torch.isfinite(loss):                               # NaNCatcher (callbacks.py:47-64)
            raise RuntimeError(f"Train loss contains a NaN at batch {step}")
        if rank == 0 and (step + 1) % log_every == 0:
            torch.cuda.synchronize()
            dt, t_last = time.time() - t_last, time.time()
            print(json.dumps({"batch": step + 1, "loss": float(loss), "lr": opt.lr * sched.factor(step),
                              "samples_per_sec": ds["train_batch_size"] * log_every / dt}), flush=True)
        if folder and save_every and (step + 1) % save_every == 0:
            trainer.consolidate()                                  # a collective under the sharded optimiser: every rank calls it
        if rank == 0 and folder and save_every and (step + 1) % save_every == 0:
            os.makedirs(folder, exist_ok=True)
            tmp = os.path.join(folder, "latest.pt.tmp")
            ema_sd = opt.ema_state_dict()
            state = {"model": {"dit." + k: v for k, v in model.dit.state_dict().items()}}
            if ema_sd is not None:                                  # the EMA weights as a loadable model state (evaluation / export)
                state["ema_model"] = {"dit." + k: v for k, v in ema_sd.items()}
            torch.save({"state": state,
                        "optimizer": opt.state_dict(), "batch": step + 1, "rng_cuda": torch.cuda.get_rng_state(),
                        "loader": loader.state_dict() if hasattr(loader, "state_dict") else None}, tmp)
            os.replace(tmp, os.path.join(folder, "latest.pt"))      # never leave a truncated latest.pt behind
    return trainer


@torch.no_grad()
def evaluate(model, eval_loader, world: int, microbatch: int = 0, opt=None) -> float:
    """Composer's eval loop around LatentDiffusion.eval_forward / DistLoss (model.py:217-229, utils.py:598-614): the EDM loss
    at eval_mask_ratio = 0 (every token kept) averaged over the eval batches of all ranks.  Like Composer, the rank batch is
    evaluated in slices of device_train_microbatch_size (at mask 0 a slice has 4x the backbone tokens of a training
    microbatch), each slice weighted by its share, and the EMA weights (once they exist) are the ones evaluated."""
    was_training = model.training
    model.eval()
    metric = model.get_metrics(is_train=False)["loss"]
    key = model.image_latents_key
    swap = opt.swap_ema() if opt is not None else None
    if swap is not None:
        swap.__enter__()
    try:
        for batch in eval_loader:
            n = batch[key].shape[0] if key in batch else next(v.shape[0] for v in batch.values() if torch.is_tensor(v))
            mb = n if microbatch <= 0 else min(microbatch, n)
            loss = None
            for s in range(0, n, mb):
                part = {k: (v[s:s + mb] if torch.is_tensor(v) and v.shape[0] == n else v) for k, v in batch.items()}
                w = min(mb, n - s) / n
                l = model.eval_forward(part)[0] * w
                loss = l if loss is None else loss + l
            model.update_metric(batch, (loss, None, None), metric)
    finally:
        if swap is not None:
            swap.__exit__(None, None, None)
        model.train(was_training)
    tot = torch.stack([torch.as_tensor(metric.loss, dtype=torch.float32, device="cuda").reshape(()),
                       torch.tensor(float(metric.batches), device="cuda")])
    if world > 1:
        dist.all_reduce(tot)                    # DistLoss: dist_reduce_fx = "sum" for both states
    return float(tot[0] / tot[1].clamp(min=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-path", required=True)
    ap.add_argument("--config-name", required=True)
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    cfg = mdcfg.load_config(a.config_path, a.config_name, a.overrides)
    if not cfg:
        raise ValueError("Config not specified. Please provide --config-path and --config-name, respectively.")
    train(cfg)


if __name__ == "__main__":
    main()
