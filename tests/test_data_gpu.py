"""Latents data path on the GPU: prefetching loader (pinned staging, copy stream, event ordering) delivers exactly the
bytes the CPU restatement reads, while the consumer overwrites / reads the batches on its own stream; and a training
step consumes loader batches directly."""
import os

import numpy as np
import pytest
import torch

from micro_diffusion_amd import data as mdata
from oracle import mds_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shards(tmp_path_factory):
    d = tmp_path_factory.mktemp("mds_gpu")
    return str(d), mds_ref.write_synthetic_latents(str(d), 64, seed=9, size_limit=1 << 22)


def test_prefetch_loader_delivers_exact_bytes(shards):
    d, samples = shards
    ds = mdata.StreamingLatentsDataset(streams=[d], shuffle=True, image_size=512, cap_seq_size=77, cap_emb_dim=1024,
                                       cap_drop_prob=0.25, batch_size=8)
    loader = mdata.LatentsLoader(ds, 8, device="cuda", rank=0, world_size=1, seed=3, depth=2, loop=True)
    it = iter(loader)
    busy = torch.randn(4096, 4096, device="cuda")
    for step in range(20):                                  # 8 batches per epoch, 2 slots: every slot is reused many times
        batch = next(it)
        epoch, b = divmod(step, 8)
        ids = loader.epoch_indices(epoch)[b * 8:(b + 1) * 8]
        assert batch["image_latents"].is_cuda and batch["image_latents"].shape == (8, 4, 64, 64)
        for _ in range(3):                                  # keep the consumer stream busy so copies overlap compute
            busy = (busy @ busy).clamp_(-1, 1)
        lat = batch["image_latents"].clone()
        cap = batch["caption_latents"].clone()
        batch["caption_latents"].mul_(batch["drop_caption_mask"].view(-1, 1, 1, 1).half())   # in-place use, like model.py
        lat, cap = lat.cpu().numpy(), cap.cpu().numpy()
        for j, i in enumerate(ids):
            assert lat[j].tobytes() == samples[i]["latents_512"], (step, j)
            assert cap[j].tobytes() == samples[i]["caption_latents"], (step, j)
        coins = loader._drop_coins(epoch, b, 8)
        assert torch.equal(batch["drop_caption_mask"].cpu(), coins)
    it.close()


def test_train_step_consumes_loader_batches(shards):
    from micro_diffusion_amd import dit as mdit
    from micro_diffusion_amd.model import LatentDiffusion, _FrozenStub
    from micro_diffusion_amd.trainer import FusedAdamW, Trainer
    from oracle import microdit_ref as orc

    d, _ = shards
    cfg = orc.tiny_config()
    net = mdit.DiT(**cfg.__dict__)
    net.load_state_dict(orc.synth_state_dict(cfg, 5))
    model = LatentDiffusion(net.to("cuda"), _FrozenStub("vae"), _FrozenStub("te"), _FrozenStub("tok"), train_mask_ratio=0.75)
    model.train()
    tr = Trainer(model, FusedAdamW(model.dit, lr=1e-4), clip_norm=0.25, microbatch_size=4)
    loader = mdata.build_streaming_latents_dataloader([d], batch_size=8, image_size=256, cap_drop_prob=0.1, shuffle=True)
    assert isinstance(loader, mdata.LatentsLoader)
    losses = [tr.train_step(batch).item() for _, batch in zip(range(3), loader)]
    assert all(np.isfinite(losses)) and all(0.05 < l < 5 for l in losses), losses


def test_train_py_eval_loop_and_ema(shards, tmp_path, capsys):
    """train.py end to end on MDS shards: the eval loop (Composer's eval_forward / DistLoss at eval_mask_ratio 0,
    model.py:217-229, utils.py:598-614, every `eval_interval`) and the EMA of the weights named by configs/res_512_*.yaml:4-9
    (smoothing s, from `ema_start` on: ema <- weights on its first batch, then s * ema + (1 - s) * weights), fused into the
    AdamW kernel."""
    import json
    import train as train_mod
    d, _ = shards
    target = "micro_diffusion.datasets.latents_loader.build_streaming_latents_dataloader"
    cfg = {
        "seed": 18,
        "model": {"_target_": "micro_diffusion.models.model.create_latent_diffusion", "dit_arch": "MicroDiT_Tiny_2", "latent_res": 32,
                  "in_channels": 4, "pos_interp_scale": 1.0, "dtype": "bfloat16", "precomputed_latents": True, "p_mean": -0.6,
                  "p_std": 1.2, "train_mask_ratio": 0.75, "vae_name": "x", "text_encoder_name": "openclip:hf-hub:apple/DFN5B-CLIP-ViT-H-14-378"},
        "optimizer": {"_target_": "torch.optim.AdamW", "lr": 1e-4, "weight_decay": 0.1, "eps": 1e-8, "betas": [0.9, 0.999]},
        "scheduler": {"_target_": "composer.optim.ConstantScheduler", "alpha": 1.0},
        "algorithms": {"gradient_clipping": {"clip_norm": 0.25, "clipping_type": "norm"}, "low_precision_layernorm": {"precision": "amp_bf16"},
                       "ema": {"_target_": "diffusion.algorithms.ema.EMA", "half_life": None, "smoothing": 0.9, "update_interval": "1ba",
                               "ema_start": "1ba"}},
        "dataset": {"image_size": 256, "train_batch_size": 8, "eval_batch_size": 8, "cap_drop_prob": 0.1,
                    "train": {"_target_": target, "datadir": [d], "drop_last": True, "shuffle": True},
                    "eval": {"_target_": target, "datadir": [d], "drop_last": False, "shuffle": False}},
        "trainer": {"max_duration": "4ba", "device_train_microbatch_size": 4, "eval_interval": "2ba", "save_interval": "0ba"},
        "misc": {"log_interval": 1},
    }
    tr = train_mod.train(cfg)
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    evals = [l for l in lines if "metrics/eval/loss" in l]
    assert [l["batch"] for l in evals] == [2, 4] and all(np.isfinite(l["metrics/eval/loss"]) and 0.05 < l["metrics/eval/loss"] < 5 for l in evals)
    opt = tr.opt
    assert opt.ema is not None and opt.ema_live
    p = tr.model.dit.flat_buffers()["p"]
    # after 4 steps with ema_start 1: ema(2) = p2, ema(3) = .9 p2 + .1 p3, ema(4) = .9 ema(3) + .1 p4 -> close to, but not equal to, p
    rel = float((opt.ema - p).norm() / p.norm())
    assert 0 < rel < 1e-2, rel
    assert tr.model.training


def test_train_py_autoresume_and_stage_handoff(tmp_path, capsys):
    """train.py checkpoints (weights under Composer's `state/model/dit.*` keys, AdamW moments keyed by parameter name, batch counter,
    loader position), `trainer.autoresume` continues the run, and a later stage picks the checkpoint up through `load_path`
    (configs/res_256_finetune.yaml:92-97: optimizer moments carried, the stage's own learning rate) -- f-3 of SURVEY.md section 8."""
    import json
    import train as train_mod
    target = "micro_diffusion.datasets.latents_loader.build_streaming_latents_dataloader"
    folder = os.path.join(tmp_path, "run")

    def cfg(max_ba, **trainer):
        return {
            "seed": 18,
            "model": {"_target_": "micro_diffusion.models.model.create_latent_diffusion", "dit_arch": "MicroDiT_Tiny_2", "latent_res": 32,
                      "in_channels": 4, "pos_interp_scale": 1.0, "dtype": "bfloat16", "precomputed_latents": True, "p_mean": -0.6,
                      "p_std": 1.2, "train_mask_ratio": 0.75, "vae_name": "x", "text_encoder_name": "openclip:hf-hub:apple/DFN5B-CLIP-ViT-H-14-378"},
            "optimizer": {"_target_": "torch.optim.AdamW", "lr": 1e-4, "weight_decay": 0.1, "eps": 1e-8, "betas": [0.9, 0.999]},
            "scheduler": {"_target_": "composer.optim.ConstantScheduler", "alpha": 1.0},
            "algorithms": {"gradient_clipping": {"clip_norm": 0.25, "clipping_type": "norm"}},
            "dataset": {"image_size": 256, "train_batch_size": 8, "eval_batch_size": 8, "cap_drop_prob": 0.1,
                        "train": {"_target_": target, "datadir": "synthetic"}},
            "trainer": dict({"max_duration": f"{max_ba}ba", "device_train_microbatch_size": 4, "save_interval": "2ba", "save_folder": folder}, **trainer),
            "misc": {"log_interval": 1},
        }
    tr = train_mod.train(cfg(2))
    ck = torch.load(os.path.join(folder, "latest.pt"), map_location="cpu")
    assert ck["batch"] == 2 and ck["optimizer"]["format"] == "by_name" and ck["optimizer"]["step"] == 2
    names = {k for k, _ in tr.model.dit.named_parameters()}
    assert set(ck["optimizer"]["m"]) == names and {k[len("dit."):] for k in ck["state"]["model"]} >= names
    m_saved = {k: v.clone() for k, v in ck["optimizer"]["m"].items()}
    capsys.readouterr()
    # ---- autoresume: two more batches on top of the two that were saved
    tr2 = train_mod.train(cfg(4, autoresume=True))
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert any(l.get("resumed_from") for l in out) and tr2.batches_seen == 4 and tr2.opt.step_count == 4
    assert [l["batch"] for l in out if "loss" in l] == [3, 4]
    # ---- next stage: weights + moments from the checkpoint of batch 2 (saved above under a different name), its own lr
    torch.save(ck, os.path.join(tmp_path, "stage1.pt"))
    c3 = cfg(1, load_path=os.path.join(tmp_path, "stage1.pt"), load_ignore_keys=["state/model/dit.pos_embed"])
    c3["trainer"]["save_folder"] = os.path.join(tmp_path, "run2")
    c3["trainer"]["save_interval"] = "0ba"
    c3["optimizer"]["lr"] = 3e-5
    tr3 = train_mod.train(c3)
    assert tr3.opt.step_count == 3 and tr3.opt.lr == 3e-5          # moments carried: the step counter continues from 2
    k = "final_layer.linear.weight"        # the reference init zeroes every gate: in the first steps only the final projection has a gradient
    b1 = 0.9
    after = tr3.opt.state_dict()["m"][k].cpu()
    # m_3 = b1 * m_2 + (1 - b1) * g_3: the carried moment is still most of it
    cos = float((after * m_saved[k]).sum() / (after.norm() * m_saved[k].norm()))
    assert cos > 0.8, cos
