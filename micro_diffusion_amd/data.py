"""Batches for the training path: `image_latents` fp16 [B,C,R,R], `caption_latents` fp16 [B,1,L,D],
`drop_caption_mask` [B] (reference micro_diffusion/datasets/latents_loader.py:43-70).

The reference streams MosaicML-MDS shards of precomputed latents; `mosaicml-streaming` is not available here and
datasets are out of scope (SURVEY.md §8f-1 marks the loader "next"), so when the MDS directories are absent the
factory returns a synthetic stream with the statistics of the real latents."""
from __future__ import annotations

import os
from typing import List, Union

import torch


class SyntheticLatents:
    """Endless iterator of device-resident synthetic batches: latents ~ N(0,1)*0.8, captions ~ N(0,1), captions
    dropped with probability cap_drop_prob (the coin of latents_loader.py:49-51)."""

    def __init__(self, batch_size: int, image_size: int = 256, cap_seq_size: int = 77, cap_emb_dim: int = 1024,
                 cap_drop_prob: float = 0.0, in_channels: int = 4, device="cuda", seed: int = 2024, length: int = 1 << 30):
        self.bs, self.res = batch_size, image_size // 8
        self.L, self.D, self.p, self.C = cap_seq_size, cap_emb_dim, cap_drop_prob, in_channels
        self.device, self.length = device, length
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.dataset = range(length)

    def __len__(self):
        return self.length // self.bs

    def __iter__(self):
        while True:
            g = self.gen
            yield {
                "image_latents": (torch.randn(self.bs, self.C, self.res, self.res, device=self.device, generator=g) * 0.8).half(),
                "caption_latents": torch.randn(self.bs, 1, self.L, self.D, device=self.device, generator=g).half(),
                "drop_caption_mask": (torch.rand(self.bs, device=self.device, generator=g) >= self.p).float(),
            }


def build_streaming_latents_dataloader(datadir: Union[str, List[str]], batch_size: int, image_size: int = 256,
                                       cap_seq_size: int = 77, cap_emb_dim: int = 1024, cap_drop_prob: float = 0.0,
                                       shuffle: bool = True, drop_last: bool = True, **dataloader_kwargs):
    """Same signature as the reference factory (latents_loader.py:73-108).  Real MDS shards need mosaicml-streaming."""
    dirs = [datadir] if isinstance(datadir, str) else list(datadir)
    have = [d for d in dirs if os.path.isdir(d)]
    if have:
        try:
            import streaming  # noqa: F401
        except ImportError as e:
            raise RuntimeError("MDS latents found but mosaicml-streaming is not installed; the MDS reader is not part "
                               "of this round (SURVEY.md §8f-1)") from e
        raise NotImplementedError("MDS shard reader: planned (SURVEY.md §8f-1)")
    rank = int(os.environ.get("RANK", "0"))
    return SyntheticLatents(batch_size, image_size, cap_seq_size, cap_emb_dim, cap_drop_prob, seed=2024 + rank)
