"""Batches for the training path: `image_latents` fp16 [B,C,R,R], `caption_latents` fp16 [B,1,L,D],
`drop_caption_mask` [B] (reference micro_diffusion/datasets/latents_loader.py:8-108).

The reference wraps `streaming.StreamingDataset` in a torch `DataLoader` (worker processes decode one sample at a time
into Python dicts, the default collate stacks them, Composer moves the batch to the device).  Here the step before the
hot path is built for one-process-per-GPU and 288 GB of HBM:

  * `StreamingLatentsDataset` keeps the reference's per-sample surface (`ds[i]` → the same dict) over the native MDS
    reader (`mds.py` / `csrc/io/mds_reader.cpp`), and adds `read_batch`, which gathers a whole batch column from the
    memory-mapped shards straight into a staging buffer (one memcpy per sample, host threads, no Python objects);
  * `LatentsLoader` replaces DataLoader + collate + device transfer: a background thread fills pinned staging slots and
    issues the H2D copies on its own HIP stream, `depth` batches ahead of the consumer; batches arrive as device tensors
    ordered against the consumer's stream with events.  The fp16→bf16 cast and the caption-drop multiply stay fused in
    the first kernel of the step (`md_cast_rows_bf16`), so the latents cross PCIe once, as fp16, and are never rewritten.

When no MDS directory exists (this image has no datasets) the factory returns `SyntheticLatents`.
Not reproduced: `streaming`'s shuffle algorithm (py1e blocks, third-party); the order here is a seeded permutation per
epoch, partitioned over ranks — the same distribution, a different sequence.
"""
from __future__ import annotations

import os
import queue
import threading
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import mds


class SyntheticLatents:
    """Iterator of device-resident synthetic batches: latents ~ N(0,1)*0.8, captions ~ N(0,1), captions dropped with
    probability cap_drop_prob (the coin of latents_loader.py:49-51).  loop=True (training): endless; loop=False (an
    evaluation set): `length` samples = len(self) batches per pass, the same batches on every pass."""

    def __init__(self, batch_size: int, image_size: int = 256, cap_seq_size: int = 77, cap_emb_dim: int = 1024,
                 cap_drop_prob: float = 0.0, in_channels: int = 4, device="cuda", seed: int = 2024, length: Optional[int] = None,
                 loop: bool = True):
        self.bs, self.res = batch_size, image_size // 8
        self.L, self.D, self.p, self.C = cap_seq_size, cap_emb_dim, cap_drop_prob, in_channels
        self.loop, self.seed = loop, seed
        if length is None:
            length = (1 << 30) if loop else 4 * batch_size
        self.device, self.length = device, length
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.dataset = range(length)

    def __len__(self):
        return self.length // self.bs

    def __iter__(self):
        if not self.loop:
            self.gen.manual_seed(self.seed)
        n = 0
        while self.loop or n < len(self):
            n += 1
            g = self.gen
            yield {
                "image_latents": (torch.randn(self.bs, self.C, self.res, self.res, device=self.device, generator=g) * 0.8).half(),
                "caption_latents": torch.randn(self.bs, 1, self.L, self.D, device=self.device, generator=g).half(),
                "drop_caption_mask": (torch.rand(self.bs, device=self.device, generator=g) >= self.p).float(),
            }


class StreamingLatentsDataset:
    """Precomputed latents in local MDS directories (latents_loader.py:8-70).

    `streams` is a list of local directories (the reference builds `Stream(remote=None, local=d)` for each,
    latents_loader.py:89); samples are indexed over the concatenation of the streams."""

    def __init__(self, streams: Optional[Sequence[str]] = None, shuffle: bool = False, image_size: Optional[int] = None,
                 cap_seq_size: Optional[int] = None, cap_emb_dim: Optional[int] = None, cap_drop_prob: float = 0.0,
                 batch_size: Optional[int] = None, **kwargs):
        if not streams:
            raise ValueError("StreamingLatentsDataset needs at least one local MDS directory")
        if image_size not in (256, 512):
            raise ValueError(f"image_size must be 256 or 512 (latents_loader.py:57,63), got {image_size}")
        self.dirs = [mds.MDSDir(getattr(s, "local", s)) for s in streams]
        self.first = np.concatenate([[0], np.cumsum([len(d) for d in self.dirs])]).astype(np.int64)
        self.shuffle, self.image_size, self.batch_size = shuffle, image_size, batch_size
        self.cap_seq_size, self.cap_emb_dim, self.cap_drop_prob = cap_seq_size, cap_emb_dim, cap_drop_prob
        self.latent_key = f"latents_{image_size}"
        self.latent_res = image_size // 8
        self._cap_col = [d.column("caption_latents") for d in self.dirs]
        self._lat_col = [d.column(self.latent_key) for d in self.dirs]
        self.cap_bytes = 2 * cap_seq_size * cap_emb_dim
        n_lat = self.dirs[0].sample_size(0, self._lat_col[0]) if len(self) else 0
        if n_lat % (2 * self.latent_res * self.latent_res):
            raise mds.MDSError(mds.SIZE_MISMATCH, f"{self.latent_key} has {n_lat} bytes: not [C,{self.latent_res},{self.latent_res}] fp16")
        self.lat_bytes = n_lat
        self.in_channels = n_lat // (2 * self.latent_res * self.latent_res)

    def __len__(self):
        return int(self.first[-1])

    def _split(self, index: int):
        if not 0 <= index < len(self):
            raise IndexError(index)
        s = int(np.searchsorted(self.first, index, side="right") - 1)
        return s, index - int(self.first[s])

    def __getitem__(self, index: int) -> Dict[str, Union[torch.Tensor, float]]:
        """The reference's per-sample dict (latents_loader.py:44-70), including the caption-drop coin from torch's
        global CPU generator."""
        s, i = self._split(int(index))
        d = self.dirs[s]
        out = {"drop_caption_mask": 0. if torch.rand(1) < self.cap_drop_prob else 1.}
        out["caption_latents"] = torch.from_numpy(
            np.frombuffer(d.read_value(i, self._cap_col[s]), dtype=np.float16).copy()).reshape(1, self.cap_seq_size, self.cap_emb_dim)
        out["image_latents"] = torch.from_numpy(
            np.frombuffer(d.read_value(i, self._lat_col[s]), dtype=np.float16).copy()).reshape(-1, self.latent_res, self.latent_res)
        return out

    def read_batch(self, indices: np.ndarray, caption_out: torch.Tensor, latents_out: torch.Tensor, n_threads: int = 4):
        """Gather samples `indices` into caller-owned CPU fp16 tensors [n,1,L,D] and [n,C,R,R] (contiguous; pinned for
        asynchronous H2D)."""
        indices = np.asarray(indices, dtype=np.int64)
        n = len(indices)
        assert caption_out.dtype == torch.float16 and latents_out.dtype == torch.float16
        assert caption_out.is_contiguous() and latents_out.is_contiguous() and caption_out.shape[0] >= n and latents_out.shape[0] >= n
        assert caption_out[0].numel() * 2 == self.cap_bytes and latents_out[0].numel() * 2 == self.lat_bytes
        if indices.size and (indices.min() < 0 or indices.max() >= len(self)):
            raise IndexError("sample index out of range")
        stream_of = np.searchsorted(self.first, indices, side="right") - 1
        for s in np.unique(stream_of):
            rows = np.nonzero(stream_of == s)[0]
            local = indices[rows] - self.first[s]
            # contiguous runs of destination rows are gathered with one call each (one run when there is one stream)
            cuts = np.nonzero(np.diff(rows) != 1)[0] + 1
            for run, loc in zip(np.split(rows, cuts), np.split(local, cuts)):
                r0 = int(run[0])
                self.dirs[s].read_batch(loc, self._cap_col[s], caption_out.data_ptr() + r0 * self.cap_bytes, self.cap_bytes,
                                        self.cap_bytes, n_threads)
                self.dirs[s].read_batch(loc, self._lat_col[s], latents_out.data_ptr() + r0 * self.lat_bytes, self.lat_bytes,
                                        self.lat_bytes, n_threads)


class _Slot:
    def __init__(self, ds: StreamingLatentsDataset, bs: int, device: torch.device):
        self.cuda = device.type == "cuda"
        shape_c = (bs, 1, ds.cap_seq_size, ds.cap_emb_dim)
        shape_l = (bs, ds.in_channels, ds.latent_res, ds.latent_res)
        self.h_cap = torch.empty(shape_c, dtype=torch.float16, pin_memory=self.cuda)
        self.h_lat = torch.empty(shape_l, dtype=torch.float16, pin_memory=self.cuda)
        self.h_drop = torch.empty(bs, dtype=torch.float32, pin_memory=self.cuda)
        if self.cuda:
            self.d_cap = torch.empty(shape_c, dtype=torch.float16, device=device)
            self.d_lat = torch.empty(shape_l, dtype=torch.float16, device=device)
            self.d_drop = torch.empty(bs, dtype=torch.float32, device=device)
            self.ready = torch.cuda.Event()        # H2D of this slot finished (recorded on the copy stream)
            self.released = torch.cuda.Event()     # consumer's work on this slot enqueued (recorded on its stream)
            self.released_valid = False
        self.n = 0


class LatentsLoader:
    """Iterable of device-resident batches; stands where the reference's `DataLoader(dataset, batch_size, drop_last)`
    stands (latents_loader.py:99-106).  `len(loader)` = batches per epoch of this rank; iteration is endless over epochs
    when `loop=True` (Composer re-iterates the DataLoader every epoch)."""

    def __init__(self, dataset: StreamingLatentsDataset, batch_size: int, drop_last: bool = True, device="cuda",
                 rank: Optional[int] = None, world_size: Optional[int] = None, seed: int = 18, depth: int = 3,
                 n_threads: int = 4, loop: bool = False):
        self.dataset, self.batch_size, self.drop_last = dataset, batch_size, drop_last
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self.seed, self.depth, self.n_threads, self.loop = seed, max(2, depth), n_threads, loop
        self.epoch = 0               # epoch the next yielded batch belongs to
        self.batch_in_epoch = 0      # batches of that epoch already yielded (resume point, see state_dict)
        per_rank = len(dataset) // self.world          # equal share per rank (tail samples rotate in with the shuffle)
        self.samples_per_rank = per_rank
        self.num_batches = per_rank // batch_size if drop_last else -(-per_rank // batch_size)
        if self.num_batches == 0:
            # with loop=True the producer would spin forever and the consumer block on an empty queue
            raise ValueError(f"dataset has {len(dataset)} samples = {per_rank} per rank, fewer than one batch of {batch_size} "
                             f"on {self.world} rank(s){' (drop_last=True)' if drop_last else ''}")

    def __len__(self):
        return self.num_batches

    def state_dict(self) -> Dict[str, int]:
        """Position of the consumer (what `streaming`'s StreamingDataset.state_dict gives Composer for a mid-epoch resume):
        the order is a pure function of (seed, epoch), so (epoch, batches consumed) is the whole state."""
        return {"epoch": self.epoch, "batch_in_epoch": self.batch_in_epoch, "seed": self.seed,
                "batch_size": self.batch_size, "world_size": self.world}

    def load_state_dict(self, sd: Dict[str, int]) -> None:
        if (sd.get("batch_size", self.batch_size), sd.get("world_size", self.world), sd.get("seed", self.seed)) != \
                (self.batch_size, self.world, self.seed):
            raise ValueError(f"loader state {sd} was saved with another batch size / world size / seed")
        self.epoch, self.batch_in_epoch = int(sd["epoch"]), int(sd["batch_in_epoch"])
        if self.num_batches and self.batch_in_epoch >= self.num_batches:
            self.epoch, self.batch_in_epoch = self.epoch + 1, 0

    def epoch_indices(self, epoch: int) -> np.ndarray:
        """Global sample ids of this rank for one epoch: a seeded permutation (identical on every rank), strided over the
        ranks.  Without shuffle: the identity order, strided the same way."""
        n = len(self.dataset)
        if self.dataset.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + epoch)
            order = torch.randperm(n, generator=g).numpy()
        else:
            order = np.arange(n, dtype=np.int64)
        return np.ascontiguousarray(order[self.rank::self.world][:self.samples_per_rank]).astype(np.int64)

    def _drop_coins(self, epoch: int, batch: int, n: int) -> torch.Tensor:
        """1 = keep the caption, 0 = drop it, P(drop) = cap_drop_prob (latents_loader.py:49-51), from a counter-based
        seed so that a run is reproducible regardless of prefetch timing."""
        g = torch.Generator().manual_seed(((self.seed * 7919 + self.rank) * 1000003 + epoch) * 1000003 + batch)
        return (torch.rand(n, generator=g) >= self.dataset.cap_drop_prob).float()

    def _produce(self, slots: List[_Slot], free_q: "queue.Queue", ready_q: "queue.Queue", stop: threading.Event):
        try:
            copy_stream = None
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
                copy_stream = torch.cuda.Stream(device=self.device)
            epoch, first_b = self.epoch, self.batch_in_epoch
            while not stop.is_set():
                idx = self.epoch_indices(epoch)
                for b in range(first_b, self.num_batches):
                    sl: _Slot = free_q.get()
                    if sl is None or stop.is_set():
                        return
                    ids = idx[b * self.batch_size:(b + 1) * self.batch_size]
                    sl.n = len(ids)
                    if sl.cuda:
                        sl.ready.synchronize()         # the previous H2D out of this slot's pinned buffers is complete
                    self.dataset.read_batch(ids, sl.h_cap, sl.h_lat, self.n_threads)
                    sl.h_drop[:sl.n] = self._drop_coins(epoch, b, sl.n)
                    if sl.cuda:
                        with torch.cuda.stream(copy_stream):
                            if sl.released_valid:
                                copy_stream.wait_event(sl.released)   # consumer kernels reading the device copy are done
                            sl.d_cap.copy_(sl.h_cap, non_blocking=True)
                            sl.d_lat.copy_(sl.h_lat, non_blocking=True)
                            sl.d_drop.copy_(sl.h_drop, non_blocking=True)
                            sl.ready.record(copy_stream)
                    ready_q.put(sl)
                epoch, first_b = epoch + 1, 0
                if not self.loop:
                    break
            ready_q.put(None)
        except BaseException as e:   # surface reader errors in the consumer thread
            ready_q.put(e)

    def __iter__(self):
        slots = [_Slot(self.dataset, self.batch_size, self.device) for _ in range(self.depth)]
        free_q: "queue.Queue" = queue.Queue()
        ready_q: "queue.Queue" = queue.Queue()
        for s in slots:
            free_q.put(s)
        stop = threading.Event()
        worker = threading.Thread(target=self._produce, args=(slots, free_q, ready_q, stop), daemon=True,
                                  name="latents-prefetch")
        worker.start()
        prev: Optional[_Slot] = None
        try:
            while True:
                if prev is not None:               # the consumer has enqueued everything that reads the previous batch
                    if prev.cuda:
                        prev.released.record(torch.cuda.current_stream(self.device))
                        prev.released_valid = True
                    free_q.put(prev)
                    prev = None
                item = ready_q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                sl: _Slot = item
                n = sl.n
                if sl.cuda:
                    torch.cuda.current_stream(self.device).wait_event(sl.ready)
                    batch = {"image_latents": sl.d_lat[:n], "caption_latents": sl.d_cap[:n], "drop_caption_mask": sl.d_drop[:n]}
                else:
                    batch = {"image_latents": sl.h_lat[:n].clone(), "caption_latents": sl.h_cap[:n].clone(),
                             "drop_caption_mask": sl.h_drop[:n].clone()}
                prev = sl
                self.batch_in_epoch += 1
                if self.batch_in_epoch >= self.num_batches:
                    self.epoch, self.batch_in_epoch = self.epoch + 1, 0
                yield batch
        finally:
            stop.set()
            free_q.put(None)


def build_streaming_latents_dataloader(datadir: Union[str, List[str]], batch_size: int, image_size: int = 256,
                                       cap_seq_size: int = 77, cap_emb_dim: int = 1024, cap_drop_prob: float = 0.0,
                                       shuffle: bool = True, drop_last: bool = True, **dataloader_kwargs):
    """Same signature as the reference factory (latents_loader.py:73-108).  DataLoader keyword arguments that configure
    worker processes (num_workers, prefetch_factor, persistent_workers, pin_memory) are accepted; `num_workers` sets the
    number of gather threads and `prefetch_factor` the number of batches in flight.

    A missing / unmounted dataset path raises (like `streaming` does): random latents are served only on explicit request,
    `datadir="synthetic"` (or `allow_synthetic=True`), which benchmarks and smoke runs use."""
    allow_synth = bool(dataloader_kwargs.pop("allow_synthetic", False))
    if isinstance(datadir, str) and datadir == "synthetic":
        allow_synth, dirs = True, []
    else:
        dirs = [datadir] if isinstance(datadir, str) else list(datadir)
    have = [d for d in dirs if os.path.isfile(os.path.join(d, "index.json"))]
    if not have and allow_synth:
        rank = int(os.environ.get("RANK", "0"))
        return SyntheticLatents(batch_size, image_size, cap_seq_size, cap_emb_dim, cap_drop_prob, seed=2024 + rank,
                                loop=bool(dataloader_kwargs.get("loop", True)))
    if len(have) != len(dirs) or not dirs:
        missing = sorted(set(dirs) - set(have))
        raise FileNotFoundError(f"MDS directories without index.json: {missing} (pass datadir='synthetic' for random latents)")
    dataset = StreamingLatentsDataset(streams=dirs, shuffle=shuffle, image_size=image_size, cap_seq_size=cap_seq_size,
                                      cap_emb_dim=cap_emb_dim, cap_drop_prob=cap_drop_prob, batch_size=batch_size)
    device = dataloader_kwargs.pop("device", "cuda" if torch.cuda.is_available() else "cpu")
    return LatentsLoader(dataset, batch_size, drop_last=drop_last, device=device,
                         n_threads=max(1, int(dataloader_kwargs.get("num_workers", 4) or 1)),
                         depth=max(2, int(dataloader_kwargs.get("prefetch_factor", 2) or 2) + 1),
                         loop=bool(dataloader_kwargs.get("loop", True)), seed=int(dataloader_kwargs.get("seed", 18)))
