"""Latents data path on the GPU: prefetching loader (pinned staging, copy stream, event ordering) delivers exactly the
bytes the CPU restatement reads, while the consumer overwrites / reads the batches on its own stream; and a training
step consumes loader batches directly."""
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
