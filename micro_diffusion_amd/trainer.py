"""Training-step runtime for the MI355X MicroDiT path: what Composer's Trainer + FSDP + GradientClipping +
torch.optim.AdamW + the LR scheduler do around `LatentDiffusion.forward` in the reference (train.py:14-123,
configs/*.yaml), re-designed for one process per GPU with RCCL over xGMI:

  * the per-rank batch (global_batch / world_size, train.py:50) is split into microbatches of
    `device_train_microbatch_size` (yaml trainer.device_train_microbatch_size); microbatch i contributes with weight
    n_i / n_rank_batch (Composer semantics): `LatentDiffusion.train_microbatch` runs forward, loss and
    the hand-written backward as one launch sequence (no autograd graph), the weight folded into md_edm_loss_train;
    gradients accumulate in the flat fp32 buffer;
  * data parallelism = the reference's FSDP SHARD_GRAD_OP (yaml trainer.fsdp_config): during the LAST microbatch's backward
    each finished segment (final layer, one DiT block, ...) of the flat gradient buffer is staged as bf16 and reduce-scattered
    asynchronously — RCCL runs on its own stream and overlaps the remaining backward kernels; every rank updates its chunk of
    the weights / moments (FusedAdamW.step_sharded, one launch over a range table) and the fresh bf16 weights are all-gathered
    bucket by bucket under the next forward (GradSync; `dp_mode="allreduce"` keeps the all-reduce form where every rank runs
    the whole optimiser pass).  Transport: torch.distributed (default) or libmicrodit_comm.so (`transport="native"`);
  * the squared gradient norm is taken per reduced bucket on a side stream behind its collective (fixed-order partial sums:
    the clip coefficient is bit-identical on all ranks, no host sync), then one fused pass clips, applies AdamW, re-emits the
    bf16 shadow weights and (all-reduce form) zeroes the gradients.
"""
from __future__ import annotations

import bisect
import collections
import os
import math
import re
import time
from ctypes import byref
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import hip


def parse_batches(v) -> int:
    """'2500ba' -> 2500 (Composer time strings; only batch units appear in configs/*.yaml)."""
    if isinstance(v, (int, float)):
        return int(v)
    m = re.fullmatch(r"\s*(\d+)\s*ba\s*", str(v))
    if not m:
        raise ValueError(f"unsupported duration '{v}' (only '<N>ba' is used by the MicroDiT configs)")
    return int(m.group(1))


class LRSchedule:
    """Composer schedulers named in configs/*.yaml, as multiplicative factors of the base LR, stepped once per batch
    (SURVEY.md Appendix C.3 — Composer source is not available offline, formulas per its documentation):
      CosineAnnealingWithWarmupScheduler(t_warmup, alpha_f), ConstantScheduler(alpha),
      ConstantWithWarmupScheduler(t_warmup, alpha)."""

    def __init__(self, kind: str, t_warmup=0, t_max=1, alpha_f: float = 0.0, alpha: float = 1.0):
        self.kind, self.t_warmup, self.t_max = kind, parse_batches(t_warmup), parse_batches(t_max)
        self.alpha_f, self.alpha = alpha_f, alpha

    @staticmethod
    def from_target(target: str, t_max, **kw) -> "LRSchedule":
        name = target.split(".")[-1]
        kinds = {"CosineAnnealingWithWarmupScheduler": "cosine_with_warmup", "ConstantScheduler": "constant",
                 "ConstantWithWarmupScheduler": "constant_with_warmup"}
        if name not in kinds:
            raise ValueError(f"scheduler {target} not supported")
        kw = {k: v for k, v in kw.items() if k in ("t_warmup", "alpha_f", "alpha")}
        return LRSchedule(kinds[name], t_max=t_max, **kw)

    def factor(self, step: int) -> float:
        """`step` = optimiser steps already taken (the first batch trains at factor(0))."""
        if self.kind == "constant":
            return self.alpha
        if self.kind == "constant_with_warmup":
            return self.alpha * min(1.0, step / self.t_warmup) if self.t_warmup > 0 else self.alpha
        if step < self.t_warmup:
            return step / self.t_warmup
        frac = min(1.0, (step - self.t_warmup) / max(1, self.t_max - self.t_warmup))
        return self.alpha_f + (1 - self.alpha_f) * 0.5 * (1 + math.cos(math.pi * frac))


class FusedAdamW:
    """torch.optim.AdamW semantics (train.py:39-43) on the DiT's flat buffers, one HIP kernel per step; optionally the
    EMA of the weights the res-512 configs name (configs/res_512_pretrain.yaml:4-9: smoothing 0.99975, every batch, from
    batch 25000 on) folded into the same pass."""

    MAX_BUCKETS = 64      # per-bucket partial sums of the gradient norm (data-parallel exchange), MD_SUMSQ_PARTIALS floats each

    def __init__(self, dit, lr: float = 2.4e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.1,
                 ema_smoothing: Optional[float] = None, ema_start: int = 0):
        self.dit = dit
        f = dit.flat_buffers()
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self.m = torch.zeros_like(f["p"])
        self.v = torch.zeros_like(f["p"])
        self.sumsq = torch.zeros(1, device=f["p"].device)
        self.partials = torch.zeros(self.MAX_BUCKETS * hip.SUMSQ_PARTIALS, device=f["p"].device)
        self.norm_slots = self.MAX_BUCKETS
        self.step_count = 0
        self.ema_smoothing, self.ema_start = ema_smoothing, int(ema_start)
        self.ema = torch.zeros_like(f["p"]) if ema_smoothing is not None else None
        self.ema_live = False
        self.last_grad_scale = 1.0

    def ensure_norm_slots(self, n: int) -> None:
        """Room for `n` per-bucket partial sums of the gradient norm (a model with more data-parallel buckets than MAX_BUCKETS:
        the sharded exchange has no other way to form ||g||)."""
        if n > self.norm_slots:
            self.partials = torch.zeros(n * hip.SUMSQ_PARTIALS, device=self.partials.device)
            self.norm_slots = n

    def _launch(self, off: int, n: int, lr: float, max_norm: float, grad_scale: float, ss, g_bf16_ptr, shadow_ptr, zero_grad: int,
                ema_mode: int) -> None:
        """md_adamw_step on elements [off, off + n) of the flat buffers; the bf16 gradient / bf16 weight output may live elsewhere
        (the packed per-rank buffers of the sharded exchange)."""
        f = self.dit.flat_buffers()
        L, st = hip.lib(), torch.cuda.current_stream().cuda_stream
        b1, b2 = self.betas
        a = hip.AdamWArgs(f["p"].data_ptr() + 4 * off, f["g"].data_ptr() + 4 * off, self.m.data_ptr() + 4 * off,
                          self.v.data_ptr() + 4 * off, shadow_ptr, ss, g_bf16_ptr,
                          (self.ema.data_ptr() + 4 * off) if self.ema is not None else None, n, lr, b1, b2, self.eps,
                          self.weight_decay, 1 - b1 ** self.step_count, 1 - b2 ** self.step_count, max_norm or 0.0, grad_scale,
                          self.ema_smoothing or 0.0, zero_grad, ema_mode)
        hip.check(L.md_adamw_step(byref(a), st), "md_adamw_step")

    def _begin_step(self, grad_scale: float) -> int:
        self.step_count += 1
        self.last_grad_scale = grad_scale
        ema_mode = 0
        if self.ema is not None and self.step_count > self.ema_start:
            ema_mode = 2 if self.ema_live else 1          # first EMA batch: ema <- weights (the EMA model starts as a copy)
            self.ema_live = True
        return ema_mode

    def step(self, lr: Optional[float] = None, max_norm: float = 0.0, grad_scale: float = 1.0, g_bf16: Optional[torch.Tensor] = None,
             norm_partials: int = 0) -> None:
        """`g_bf16`: take the gradients from this bf16 flat buffer (data-parallel exchange buffer) instead of the fp32
        accumulators.  `norm_partials` > 0: that many per-bucket partial sums of squares are already in self.partials."""
        f = self.dit.flat_buffers()
        L, st = hip.lib(), torch.cuda.current_stream().cuda_stream
        ema_mode = self._begin_step(grad_scale)
        ss = None
        if max_norm and max_norm > 0:
            if norm_partials <= 0:
                src = g_bf16 if g_bf16 is not None else f["g"]
                hip.check(L.md_sumsq(src.data_ptr(), 1 if g_bf16 is not None else 0, f["total"], self.partials.data_ptr(), st), "md_sumsq")
                norm_partials = hip.SUMSQ_PARTIALS
            hip.check(L.md_sumsq_finish(self.partials.data_ptr(), norm_partials, self.sumsq.data_ptr(), st), "md_sumsq_finish")
            ss = self.sumsq.data_ptr()
        self._launch(0, f["total"], self.lr if lr is None else lr, max_norm, grad_scale, ss,
                     g_bf16.data_ptr() if g_bf16 is not None else None, f["s"].data_ptr(), 1, ema_mode)
        self.dit.mark_shadow_fresh()

    def step_sharded(self, sync: "GradSync", lr: Optional[float] = None, max_norm: float = 0.0, grad_scale: float = 1.0,
                     norm_slots: int = 0, chunk_of: Optional[int] = None) -> None:
        """The rank's share of the step under GradSync(mode='sharded'): ||g||^2 from the ranks' chunk norms (one scalar
        all-reduce), AdamW on this rank's chunk of every matrix-shaped bucket (gradient from the reduce-scattered bf16 buffer,
        fresh bf16 weights into the packed send buffer) and on the whole "small" region (every rank), then the asynchronous
        all-gather of the bf16 weights.  The fp32 accumulators were cleared by the staging cast.
        `chunk_of` (measurement aid, single rank): pretend to be one of `chunk_of` ranks — update only the first 1 / chunk_of of
        every bucket — to time the per-rank optimiser pass of an N-GPU run on one GPU; the model is NOT valid afterwards."""
        f = self.dit.flat_buffers()
        ema_mode = self._begin_step(grad_scale)
        lr = self.lr if lr is None else lr
        ss = None
        if max_norm and max_norm > 0:
            sync.finish_sharded_norm(norm_slots, self.sumsq)
            ss = self.sumsq.data_ptr()
        # ONE launch over the rank's chunks (md_adamw_step_ranges: the kernel walks the packed space the reduce-scattered gradient
        # and the send buffer live in, a range table maps it onto the flat masters / moments)
        ranges = [(lo + sync.rank * chunk, chunk if chunk_of is None else (hi - lo) // chunk_of // 64 * 64) for _, lo, hi, chunk, _ in sync.plan]
        ranges = [r for r in ranges if r[1] > 0]
        if len(ranges) <= 64:        # (with chunk_of the packed gradient / weight offsets no longer line up: timing only, as documented)
            import ctypes
            key = tuple(ranges)
            if getattr(self, "_range_key", None) != key:
                self._range_key = key
                self._range_off = (ctypes.c_int64 * len(ranges))(*[r[0] for r in ranges])
                self._range_cnt = (ctypes.c_int64 * len(ranges))(*[r[1] for r in ranges])
            b1, b2 = self.betas
            a = hip.AdamWArgs(f["p"].data_ptr(), f["g"].data_ptr(), self.m.data_ptr(), self.v.data_ptr(), sync.ssend.data_ptr(), ss,
                              sync.gred.data_ptr(), self.ema.data_ptr() if self.ema is not None else None, 0, lr, b1, b2, self.eps,
                              self.weight_decay, 1 - b1 ** self.step_count, 1 - b2 ** self.step_count, max_norm or 0.0, grad_scale,
                              self.ema_smoothing or 0.0, 0, ema_mode)
            hip.check(hip.lib().md_adamw_step_ranges(byref(a), self._range_off, self._range_cnt, len(ranges),
                                                     torch.cuda.current_stream().cuda_stream), "md_adamw_step_ranges")
        else:                                   # more buckets than the kernel's table holds: chunk by chunk
            for _, lo, hi, chunk, olo in sync.plan:
                self._launch(lo + sync.rank * chunk, chunk, lr, max_norm, grad_scale, ss, sync.gred.data_ptr() + 2 * olo,
                             sync.ssend.data_ptr() + 2 * olo, 0, ema_mode)
        if sync.small is not None:
            lo, hi = sync.small
            self._launch(lo, hi - lo, lr, max_norm, grad_scale, ss, sync.gbf.data_ptr() + 2 * lo, f["s"].data_ptr() + 2 * lo, 0, ema_mode)
        sync.gather_shadows()
        self.dit.mark_shadow_fresh()

    def grad_norm(self) -> torch.Tensor:
        """||g||_2 of the last step's (rank-averaged) gradient before clipping (device scalar; valid after step() with
        max_norm > 0).  The sum of squares is taken over the rank-SUMMED gradient; the 1 / world factor is applied here as in
        the optimiser kernel."""
        return self.sumsq.sqrt() * self.last_grad_scale

    def _by_name(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        f = self.dit.flat_buffers()
        return {name: flat[f["offs"][name]:f["offs"][name] + view.numel()].view(view.shape) for name, view in f["P"].items()}

    def state_dict(self):
        """Moments (and EMA) keyed by parameter name, like torch.optim's per-parameter state: independent of the flat layout."""
        sd = {"m": self._by_name(self.m), "v": self._by_name(self.v), "step": self.step_count, "format": "by_name"}
        if self.ema is not None:
            sd["ema"], sd["ema_live"] = self._by_name(self.ema), self.ema_live
        return sd

    def load_state_dict(self, sd):
        def put(dst_flat, src):
            if isinstance(src, dict):
                views = self._by_name(dst_flat)
                if set(views) != set(src):
                    raise RuntimeError(f"optimizer state names differ from the model's: {sorted(set(views) ^ set(src))[:4]} ...")
                for k, v in views.items():
                    v.copy_(src[k])
            else:                           # a flat tensor is only meaningful under the flat layout it was saved from, and the
                # layout has changed between rounds (dit.flat_layout: the adaLN weights moved to the front) -- a matching element
                # count proves nothing, so flat state is refused
                raise RuntimeError("flat-format optimizer state is not accepted: re-save it keyed by parameter name "
                                   "(FusedAdamW.state_dict(), format 'by_name')")
        put(self.m, sd["m"])
        put(self.v, sd["v"])
        self.step_count = int(sd["step"])
        if self.ema is not None and "ema" in sd:
            put(self.ema, sd["ema"])
            self.ema_live = bool(sd.get("ema_live", True))
        elif self.ema is not None and self.step_count > self.ema_start:
            # resuming past ema_start from a checkpoint that holds no EMA: the average would silently restart from the
            # current weights.  That is what Composer's EMA does when it is first enabled, but it must not go unnoticed.
            import warnings
            warnings.warn(f"optimizer state at step {self.step_count} (> ema_start {self.ema_start}) carries no EMA weights: "
                          "the EMA restarts from the current weights")
        elif self.ema is None and "ema" in sd:
            import warnings
            warnings.warn("the checkpoint carries EMA weights but this run configures no EMA: they are dropped")

    # ------------------------------------------------------------------ EMA weights for evaluation / export
    def ema_state_dict(self):
        """The EMA weights as a model state_dict (same keys / shapes as dit.state_dict() for parameters), or None before the
        first EMA batch.  Saved by train.py next to the raw weights (Composer's EMA algorithm keeps them in its own state)."""
        if self.ema is None or not self.ema_live:
            return None
        f = self.dit.flat_buffers()
        out = {}
        for name, view in f["P"].items():
            o = f["offs"][name]
            out[name] = self.ema[o:o + view.numel()].view(view.shape)
        return out

    class _EmaSwap:
        def __init__(self, opt):
            self.opt = opt

        def __enter__(self):
            o = self.opt
            self.active = o.ema is not None and o.ema_live
            if self.active and getattr(o.dit, "shadow_is_authoritative", False):
                # sharded optimiser: the fp32 masters of the other ranks' chunks are stale and refresh_shadow() would refuse AFTER
                # the swap, leaving masters and EMA exchanged -- refuse here, before anything is touched
                raise RuntimeError("swap_ema() with stale fp32 masters (sharded optimiser): call Trainer.consolidate() first")
            if self.active:                       # off the step path: plain tensor swaps, then re-derive the bf16 shadow
                f = o.dit.flat_buffers()
                tmp = f["p"].clone()
                f["p"].copy_(o.ema)
                o.ema.copy_(tmp)
                o.dit.refresh_shadow(force=True)
            return self.active

        def __exit__(self, *exc):
            if self.active:
                o = self.opt
                f = o.dit.flat_buffers()
                tmp = f["p"].clone()
                f["p"].copy_(o.ema)
                o.ema.copy_(tmp)
                o.dit.refresh_shadow(force=True)
            return False

    def swap_ema(self):
        """Context manager: the model computes with the EMA weights inside (Composer's EMA swaps them in for evaluation);
        a no-op before the first EMA batch."""
        return FusedAdamW._EmaSwap(self)


RCCL_CHANNELS_DEFAULT = 8      # CUs handed to RCCL per rank: 8 channels move the 2 x 2 GB of a step at well over the ~30 GB/s the
#                                overlap needs (xGMI: 7 links x ~50 GB/s per direction); every channel costs the GEMMs 1 / 256


def cap_rccl_channels(n: int = RCCL_CHANNELS_DEFAULT) -> int:
    """Call BEFORE the process group / communicator is created: bounds the CUs RCCL's persistent kernels occupy
    (NCCL_MAX_NCHANNELS; a value the user exported wins).  Returns the cap in force."""
    import os
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
    cap = rccl_channel_cap()          # tolerant parse: a non-numeric exported value counts as "no cap" (0)
    if cap > 0:
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(n, cap)))
    return cap


def rccl_channel_cap() -> int:
    import os
    try:
        return max(0, min(64, int(os.environ.get("NCCL_MAX_NCHANNELS", "0"))))
    except ValueError:
        return 0


def shard_plan(buckets, world: int):
    """Rank chunks of the data-parallel buckets.  `buckets` = [(key, lo, hi)] tiling the flat buffer (dit.bucket_ranges; every
    boundary is a multiple of dit._BUCKET_ALIGN, so (hi - lo) / world is a whole number of 128-byte lines for world <= 16).
    Returns [(key, lo, hi, chunk, olo)] for the matrix-shaped buckets in flat (= forward) order — rank r owns
    [lo + r * chunk, lo + (r + 1) * chunk) of bucket i and keeps its reduced gradient / fresh bf16 weights at [olo, olo + chunk)
    of a packed per-rank buffer — and the (lo, hi) of the "small" region every rank owns whole."""
    plan, olo, small = [], 0, None
    for key, lo, hi in buckets:
        if key == "small":
            small = (lo, hi)
            continue
        n = hi - lo
        if n % world or (n // world) % 64:
            raise ValueError(f"bucket {key} [{lo}, {hi}) does not split into {world} aligned chunks")
        plan.append((key, lo, hi, n // world, olo))
        olo += n // world
    return plan, small, olo


class GradSync:
    """Overlapped data-parallel gradient exchange (the reference's FSDP gradient reduction, configs/*.yaml fsdp_config:
    SHARD_GRAD_OP = gradients and optimiser state sharded, weights whole).

    As soon as the engine reports the backward of a segment (final layer, one DiT block, ...) as enqueued, that bucket of the
    flat gradient buffer is handed to RCCL (torch.distributed, backend "nccl" = RCCL over xGMI; async: the collective runs on
    RCCL's own stream behind an event on the compute stream) while the remaining backward kernels keep the CUs busy.

    mode "sharded" (default for N > 1 with the bf16 exchange): ZeRO-1 in the reference's sense —
      * a bucket is staged as bf16 (md_cast_f32_bf16_clear: the fp32 accumulators come back zeroed) and REDUCE-SCATTERED: rank r
        receives the sum of chunk r only (half the bytes an all-reduce moves per rank);
      * each rank takes the squared norm of ITS chunks on a side stream; one scalar all-reduce completes ||g||^2 (identical on
        all ranks);
      * FusedAdamW.step_sharded updates the rank's chunks of weights / moments (1 / N of the 34 B per parameter pass) and writes
        their fresh bf16 values into a packed send buffer;
      * the bf16 shadow weights are ALL-GATHERED bucket by bucket in forward order, asynchronously; the engine waits for a
        bucket right before the first kernel that reads it (DiTEngine.before_segment), so the gather hides under the next forward;
      * one-dimensional tensors (the "small" region: the engine reads them from the fp32 masters) are all-reduced as one bucket
        and updated by every rank.
      The fp32 masters / moments of foreign chunks go stale: Trainer.consolidate() all-gathers them before a checkpoint or an
      evaluation on EMA weights.
    mode "allreduce": every bucket all-reduced, every rank runs the whole optimiser pass (round 2's exchange; also the fp32 /
      gloo parity path).  Exchange formats: "bf16" (2.33 GB per step for XL/2; FSDP's default mixed precision reduces in the low
      precision too, SURVEY.md C.7 -- measured effect at 8 ranks: tests/test_abi_and_dp_cpu.py) or "fp32" in place.
    The squared norm of every reduced bucket is taken on a side stream right behind its collective (deterministic partial sums,
    md_sumsq), so the clip coefficient needs no extra pass after the last bucket and is bit-identical on all ranks."""

    def __init__(self, dit, process_group=None, exchange: str = "auto", single_rank_exchange: bool = False, mode: str = "auto",
                 transport: str = "torch"):
        """`single_rank_exchange`: run the whole exchange path (staging cast, asynchronous collectives, side-stream norm) also on
        a process group of ONE rank, where every collective is the identity — the only way to drive the RCCL code path on a box
        with a single GPU (tests/test_dp_gpu.py); never set in production.
        `transport`: "torch" — the collectives are torch.distributed calls (backend "nccl" = RCCL); "native" — they go through
        libmicrodit_comm.so (include/microdit_comm.h: md_comm_*_bucket on the communicator's own high-priority HIP stream), and
        torch.distributed (any backend) only carries the 128-byte unique id at start-up."""
        self.dit = dit
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.enabled = self.world > 1 or (single_rank_exchange and dist.is_initialized())
        assert transport in ("torch", "native")
        self.transport = transport if self.enabled else "torch"
        self.comm = None
        if exchange == "auto":
            exchange = "bf16" if (self.enabled and (dist.get_backend(process_group) == "nccl" or self.transport == "native")) else "fp32"
        assert exchange in ("bf16", "fp32")
        f = dit.flat_buffers()
        buckets = f.get("buckets")
        if buckets is None:                                   # a bare table (CPU tests): derive the ranges here
            from .dit import bucket_ranges
            buckets = bucket_ranges(dit._table, f["offs"], f["total"])
        self.bucket_list = list(buckets)
        self.mode_note = None
        if mode == "auto":
            mode = "sharded" if (self.enabled and exchange == "bf16") else "allreduce"
            if mode == "sharded":
                # buckets are multiples of dit._BUCKET_ALIGN = 1024 elements: they split into 64-element-aligned rank chunks when
                # the world size divides 16 (2, 4, 8, 16).  Any other world size (3, 5, 6, 7 GPUs) keeps the all-reduce exchange,
                # which has no such constraint; only an EXPLICIT mode="sharded" raises.
                try:
                    shard_plan(self.bucket_list, self.world)
                except ValueError as e:
                    mode = "allreduce"
                    self.mode_note = f"world size {self.world}: {e}; falling back to the all-reduce exchange"
                    import warnings
                    warnings.warn("GradSync: " + self.mode_note)
        assert mode in ("sharded", "allreduce")
        if mode == "sharded" and exchange != "bf16":
            raise ValueError("the sharded exchange stages gradients as bf16 (exchange='bf16')")
        self.exchange, self.mode = exchange, mode
        # segment name reported by the engine -> its ranges ("rest" also flushes "small": both complete with the last segment)
        self.ranges: Dict[str, List[tuple]] = {}
        for key, lo, hi in self.bucket_list:
            self.ranges.setdefault(key, []).append((lo, hi))
        self.pending = []
        self._adaln_b_sent = False           # the backbone's adaLN bucket went out with "blocks.0" in this backward
        self._wire = collections.deque()     # every collective handle in issue order, for in_flight()
        # in_flight() decides md_gemm_args.cu_limit, and the CU count decides whether a bf16-output GEMM takes the split-K tail form
        # (another fp32 summation order): a LIVE completion poll therefore makes a data-parallel run not bit-reproducible from run
        # to run (replicas stay identical: gradients are reduced).  MD_DP_DETERMINISTIC=1 answers from host state instead (issued
        # and not yet consumed): reproducible, the grids give up their CUs for longer.
        self.deterministic = os.environ.get("MD_DP_DETERMINISTIC", "0") == "1"
        self.store_bf16 = os.environ.get("MD_DP_STORE_BF16", "1") != "0"    # one-microbatch steps: begin_backward() (A/B: 0 = off)
        self.last_stored = 0
        self.active = False
        self.buckets = 0                 # buckets handed over in the current step (= partial-sum slots in use)
        self.last_buckets = 0            # ... in the last finished step (bench.py dp block)
        self.step_bytes = self.last_bytes = 0   # bytes handed to the collectives in the current / last step
        self.norm_partials: Optional[torch.Tensor] = None     # set by the Trainer: FusedAdamW.partials
        on_gpu = f["g"].is_cuda
        dev = f["g"].device
        self.gbf = torch.zeros(f["total"], device=dev, dtype=torch.bfloat16) if (exchange == "bf16" and self.enabled) else None
        self.side = torch.cuda.Stream(device=dev) if (self.enabled and on_gpu) else None
        if self.transport == "native":
            if not on_gpu:
                raise ValueError("the native transport moves device buffers over RCCL")
            from . import comm as mdcomm
            self.comm = mdcomm.Comm.from_torch_distributed(process_group)
        # torch.distributed calls on GPU tensors need RCCL underneath; under gloo they bounce through the host
        self.torch_bounce = self.enabled and on_gpu and dist.get_backend(process_group) != "nccl"
        self.host_bounce = self.torch_bounce and self.comm is None
        self.plan = self.small = None
        self.gather_work: Dict[str, list] = {}
        if mode == "sharded":
            self.plan, self.small, own = shard_plan(self.bucket_list, self.world)
            self.by_range = {(lo, hi): (chunk, olo) for _, lo, hi, chunk, olo in self.plan}
            self.gred = torch.zeros(max(own, 8), device=dev, dtype=torch.bfloat16)     # reduced gradient of this rank's chunks
            self.ssend = torch.zeros(max(own, 8), device=dev, dtype=torch.bfloat16)    # fresh bf16 weights of this rank's chunks
            # [small-region partial sums (SUMSQ_PARTIALS floats) | sum over the ranks' chunk norms (1 float)] -> md_sumsq_finish
            self.fin = torch.zeros(hip.SUMSQ_PARTIALS + 8, device=dev)

    def norm_slots(self) -> int:
        return 0 if self.norm_partials is None else self.norm_partials.numel() // hip.SUMSQ_PARTIALS

    def _track(self, work):
        if work is not None:
            self._wire.append(work)
        return work

    def in_flight(self) -> bool:
        """A collective of this exchange may be resident on the GPU right now.  Every collective handle is also kept in issue
        order in `_wire`; handles that report completion are dropped from its front (one query per call in the steady state:
        collectives of one communicator finish in order), so the GEMM grids get their CUs back as soon as the wire is idle --
        not only when finish() / wait_gather() has consumed the handles."""
        if self.deterministic:         # host state only: issued and not yet consumed by finish() / wait_gather()
            return bool(self.pending) or bool(self.gather_work)
        w = self._wire
        while w:
            q = getattr(w[0], "is_completed", None)
            try:
                if q is None or not q():
                    return True
            except Exception:       # a handle that cannot be queried counts as busy
                return True
            w.popleft()
        return False

    def describe(self) -> str:
        if not self.enabled:
            return "none (single rank)"
        n = len(self.bucket_list)
        via = " [md_comm over RCCL]" if self.comm is not None else ""
        if self.mode == "sharded":
            return (f"bf16 reduce-scatter per backward segment ({n} buckets, 1-D tensors all-reduced), sharded AdamW, bf16 weights "
                    f"all-gathered under the next forward{via}")
        return f"{self.exchange} all-reduce per backward segment ({n} buckets), overlapped with backward{via}"

    # ------------------------------------------------------------------ collectives (RCCL, or a host bounce under gloo)
    def _all_reduce(self, buf):
        if self.comm is not None:
            return self.comm.all_reduce(buf)
        if self.host_bounce:
            h = buf.float().cpu() if buf.dtype == torch.bfloat16 else buf.cpu()
            dist.all_reduce(h, group=self.pg)
            buf.copy_(h.to(buf.dtype))
            return None
        return dist.all_reduce(buf, group=self.pg, async_op=True)

    def _reduce_scatter(self, out, buf):
        if self.comm is not None:
            return self.comm.reduce_scatter(out, buf)
        if self.host_bounce:               # gloo has no reduce-scatter: all-reduce on the host, keep this rank's chunk
            h = buf.float().cpu()
            dist.all_reduce(h, group=self.pg)
            c = out.numel()
            out.copy_(h[self.rank * c:(self.rank + 1) * c].to(out.dtype))
            return None
        return dist.reduce_scatter_tensor(out, buf, group=self.pg, async_op=True)

    def _all_gather(self, out, mine):
        if self.comm is not None:
            return self.comm.all_gather(out, mine)
        if self.host_bounce:
            h = mine.float().cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.pg)
            out.copy_(torch.cat(parts).to(out.dtype))
            return None
        return dist.all_gather_into_tensor(out, mine, group=self.pg, async_op=True)

    def _norm_after(self, work, buf, is_bf16, slot_ptr):
        """||buf||^2 partial sums on the side stream, right behind the collective that produced buf."""
        if self.side is None:
            return
        if work is None:
            self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            if work is not None:
                work.wait()              # stream-side dependency under NCCL; the norm overlaps the remaining backward
            hip.check(hip.lib().md_sumsq(buf.data_ptr(), 1 if is_bf16 else 0, buf.numel(), slot_ptr, self.side.cuda_stream), "md_sumsq")

    # ------------------------------------------------------------------ one-microbatch steps: weight gradients stored as bf16
    def begin_backward(self, single_microbatch: bool) -> None:
        """Called by the Trainer in front of the step's LAST microbatch.  When that microbatch is also the first (a rank of the
        8-GPU run: device_train_microbatch_size 256 = the rank's batch, configs/res_256_pretrain.yaml:24,111) and the exchange is
        the sharded bf16 one, the engine stores every weight gradient straight into `gbf` (DiTEngine.wgrad_bf16): the fp32
        accumulators of those tensors are neither read nor written in this step and _exchange skips their cast + clear."""
        eng = self.dit.engine
        eng.wgrad_bf16 = None
        if not (single_microbatch and self.store_bf16 and self.enabled and self.mode == "sharded" and self.gbf is not None and self.gbf.is_cuda):
            return
        f = self.dit.flat_buffers()
        eng.wgrad_bf16 = {"g_lo": f["g"].data_ptr(), "g_hi": f["g"].data_ptr() + 4 * f["g"].numel(), "gbf": self.gbf.data_ptr(),
                          "written": {}}

    def end_backward(self) -> None:
        t = self.dit.engine.wgrad_bf16
        self.last_stored = len(t["written"]) if t is not None else 0      # gradient launches that went straight to the exchange buffer
        self.dit.engine.wgrad_bf16 = None

    def _tensors_of(self, lo: int, hi: int):
        """(flat offset, numel) of the parameters inside the range, in address order (cached)."""
        c = getattr(self, "_range_tensors", None)
        if c is None:
            c = self._range_tensors = {}
        if (lo, hi) not in c:
            f = self.dit.flat_buffers()
            c[(lo, hi)] = sorted((o, f["P"][n].numel()) for n, o in f["offs"].items() if n in f["P"] and lo <= o < hi)
        return c[(lo, hi)]

    def _exchange(self, lo: int, hi: int) -> None:
        f = self.dit.flat_buffers()
        g = f["g"][lo:hi]
        st = torch.cuda.current_stream().cuda_stream if g.is_cuda else None
        sharded = self.mode == "sharded"
        small = sharded and self.small is not None and (lo, hi) == tuple(self.small)
        stored = self.dit.engine.wgrad_bf16 if sharded else None
        if self.exchange == "bf16":
            buf = self.gbf[lo:hi]
            if stored is not None:
                # one-microbatch step: tensors the engine stored as bf16 are in `buf` already; what it did not store (one-dimensional
                # tensors, a tensor without a gradient this step, an fp32 fall-back) is cast + cleared tensor by tensor, runs merged
                # (alignment gaps between two such tensors are zero in both buffers: a run covers them)
                g0, runs, cur = f["g"].data_ptr(), [], None
                spans = sorted(((a - g0) // 4, (a - g0) // 4 + cnt) for a, cnt in stored["written"].items())
                starts = [a for a, _ in spans]
                for o, n in self._tensors_of(lo, hi):
                    i = bisect.bisect_right(starts, o) - 1
                    if i >= 0 and o + n <= spans[i][1]:            # inside a stored span (a span may cover adjacent tensors: [w1; w2])
                        cur = None
                    elif cur is None:
                        cur = [o, o + n]
                        runs.append(cur)
                    else:
                        cur[1] = o + n
                for a, b in runs:
                    hip.check(hip.lib().md_cast_f32_bf16_clear(g0 + 4 * a, self.gbf.data_ptr() + 2 * a, b - a, st), "cast_clear")
            elif sharded:                 # the fp32 accumulators come back cleared (no later pass visits all of them)
                hip.check(hip.lib().md_cast_f32_bf16_clear(g.data_ptr(), buf.data_ptr(), hi - lo, st), "cast_clear")
            else:
                hip.check(hip.lib().md_cast_f32_bf16(g.data_ptr(), buf.data_ptr(), hi - lo, None, st), "cast")
        else:
            buf = g
        self.step_bytes += buf.numel() * buf.element_size()
        if sharded and not small:
            chunk, olo = self.by_range[(lo, hi)]
            red = self.gred[olo:olo + chunk]
            work = self._reduce_scatter(red, buf)
            slot = self.buckets
            self.buckets += 1
            if self.norm_partials is not None:
                if slot >= self.norm_slots():
                    raise RuntimeError(f"{slot + 1} reduce-scattered buckets but {self.norm_slots()} norm slots: size them with "
                                       "FusedAdamW.ensure_norm_slots(len(bucket_list)) (the Trainer does)")
                self._norm_after(work, red, True, self.norm_partials.data_ptr() + 4 * slot * hip.SUMSQ_PARTIALS)
        elif small:
            work = self._all_reduce(buf)
            self._norm_after(work, buf, True, self.fin.data_ptr())        # identical on every rank: added once, after the scalar all-reduce
        else:
            work = self._all_reduce(buf)
            slot = self.buckets
            self.buckets += 1
            if self.norm_partials is not None and slot < self.norm_slots():
                self._norm_after(work, buf, self.exchange == "bf16", self.norm_partials.data_ptr() + 4 * slot * hip.SUMSQ_PARTIALS)
        if work is not None:
            self.pending.append(self._track(work))

    def on_segment(self, name: str) -> None:
        if not self.active or not self.enabled:
            return
        for lo, hi in self.ranges.get(name, []):
            self._exchange(lo, hi)
        # The modulation weights of all blocks live in two buckets of their own (arch.bucket_key): the backbone's part is complete
        # once the backward of the FIRST backbone block is enqueued (409 MB of the 2.33 GB at XL/2: handed over here, the mixer's
        # backward still covers it), the mixer's part (57 MB) and the one-dimensional tensors with the last segment.
        if name == "blocks.0" and not self._adaln_b_sent:
            self._adaln_b_sent = True
            for lo, hi in self.ranges.get("adaln.b", []):
                self._exchange(lo, hi)
        if name == "rest":
            for key in (() if self._adaln_b_sent else ("adaln.b",)) + ("adaln.m", "small"):
                for lo, hi in self.ranges.get(key, []):
                    self._exchange(lo, hi)
            self._adaln_b_sent = False

    def finish(self) -> int:
        """Wait for every bucket; returns the number of norm partial-sum slots filled (0 = the optimiser takes the norm itself;
        sharded mode: the finished ||g||^2 is written to `sumsq_out` by finish_sharded_norm instead)."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if not self.gather_work:
            self._wire.clear()           # everything issued has been waited for (in_flight() may never be polled: world 1, gloo, cap 0)
        n = self.buckets if (self.norm_partials is not None and self.side is not None and 0 < self.buckets <= self.norm_slots()) else 0
        if self.side is not None and self.buckets:
            torch.cuda.current_stream().wait_stream(self.side)
        self.last_buckets, self.last_bytes = self.buckets, self.step_bytes
        self.buckets = self.step_bytes = 0
        return n

    def finish_sharded_norm(self, slots: int, sumsq_out: torch.Tensor) -> None:
        """||g||^2 of the rank-summed gradient from the ranks' chunk partial sums: local finish -> scalar all-reduce -> + the small
        region's partial sums (every rank holds the same ones) -> sumsq_out.  Fixed summation order everywhere: identical bits on
        all ranks."""
        L, st = hip.lib(), torch.cuda.current_stream().cuda_stream
        P = hip.SUMSQ_PARTIALS
        tot = self.fin[P:P + 1]
        hip.check(L.md_sumsq_finish(self.norm_partials.data_ptr(), slots * P, tot.data_ptr(), st), "md_sumsq_finish")
        w = self._all_reduce(tot)
        if w is not None:
            w.wait()
        hip.check(L.md_sumsq_finish(self.fin.data_ptr(), P + 1, sumsq_out.data_ptr(), st), "md_sumsq_finish")

    def gather_shadows(self) -> None:
        """All-gather the fresh bf16 weights bucket by bucket in forward order (asynchronous); wait_gather(key) is the engine's
        hook in front of the first kernel that reads a bucket."""
        s = self.dit.flat_buffers()["s"]
        for key, lo, hi, chunk, olo in self.plan:
            w = self._all_gather(s[lo:hi], self.ssend[olo:olo + chunk])
            self.step_bytes += (hi - lo) * 2
            if w is not None:
                self.gather_work.setdefault(key, []).append(self._track(w))

    def wait_gather(self, key: Optional[str] = None) -> None:
        if not self.gather_work:
            return
        keys = [key] if key is not None else list(self.gather_work)
        for k in keys:
            for w in self.gather_work.pop(k, []):
                w.wait()
        if not self.gather_work and not self.pending:
            self._wire.clear()


class Trainer:
    def __init__(self, model, optimizer: FusedAdamW, schedule: Optional[LRSchedule] = None, clip_norm: float = 0.0,
                 microbatch_size: int = 256, process_group=None, log: Optional[Callable[[dict], None]] = None,
                 exchange: str = "auto", single_rank_exchange: bool = False, dp_mode: str = "auto", transport: str = "auto"):
        """dp_mode: "sharded" (reduce-scatter + sharded AdamW + all-gather of the bf16 weights, the reference's SHARD_GRAD_OP;
        default for N > 1 over RCCL) or "allreduce" (every rank runs the whole optimiser pass); "auto" also honours the
        MD_DP_MODE environment variable."""
        import os
        self.model, self.opt, self.schedule, self.clip_norm = model, optimizer, schedule, clip_norm
        self.microbatch_size = microbatch_size
        if dp_mode == "auto" and os.environ.get("MD_DP_MODE"):
            dp_mode = os.environ["MD_DP_MODE"]
        if exchange == "auto" and os.environ.get("MD_DP_EXCHANGE"):
            exchange = os.environ["MD_DP_EXCHANGE"]
        if transport == "auto":                # MD_COMM=native: the exchange through libmicrodit_comm.so instead of torch.distributed
            transport = os.environ.get("MD_COMM", "torch")
        model.dit._ensure_flat()
        self.sync = GradSync(model.dit, process_group, exchange=exchange, single_rank_exchange=single_rank_exchange, mode=dp_mode,
                             transport=transport)
        optimizer.ensure_norm_slots(len(self.sync.bucket_list))     # one slot per bucket: the sharded norm has no other source
        self.sync.norm_partials = optimizer.partials
        self.world = self.sync.world
        self.sharded = self.sync.enabled and self.sync.mode == "sharded"
        self.stale_foreign_chunks = False    # sharded: fp32 masters / moments of the other ranks' chunks are out of date
        self.shard_chunk_of = None           # measurement aid, see FusedAdamW.step_sharded
        model.dit._on_segment = self.sync.on_segment
        # RCCL's kernels hold one CU per channel while a collective runs and a workgroup of the persistent GEMM needs a whole CU:
        # leave those CUs out of the GEMM grids for as long as a collective is in flight (md_gemm_args.cu_limit; the channel
        # count is capped by NCCL_MAX_NCHANNELS, which cap_rccl_channels() sets before the process group is created)
        # (a gloo / host-bounce exchange runs no RCCL kernel: nothing holds a CU, the grids stay whole)
        on_rccl = self.sync.comm is not None or not getattr(self.sync, "host_bounce", False)
        self.rccl_channels = rccl_channel_cap() if (self.sync.enabled and self.world > 1 and on_rccl) else 0
        if self.rccl_channels:
            lim = hip.NUM_CU - self.rccl_channels
            model.dit.engine.cu_limit_fn = lambda s=self.sync, lim=lim: (lim if s.in_flight() else 0)
        model.dit.engine.before_segment = self.sync.wait_gather if self.sharded else None
        # the Trainer runs forward -> backward strictly in turn: activations may live in the engine's fixed-address arenas
        model.dit._ensure_flat()
        model.dit.engine.use_arena = True
        self.batches_seen = 0
        self.log = log
        self._win: List[tuple] = []
        self.measure_comm = False          # bench.py: event pairs around GradSync.finish() (exposed exchange time)
        self._comm_events: List[tuple] = []
        self._opt_events: List[tuple] = []

    def train_step(self, batch: dict) -> torch.Tensor:
        """One optimisation step on this rank's share of the global batch.  Returns the rank-mean loss (device scalar).
        Microbatch i contributes with weight n_i / n (Composer scales each microbatch loss by its share of the rank batch before
        backward): the weight goes into md_edm_loss_train, which pre-multiplies dL/dF and accumulates the
        weighted loss on the device — no autograd graph and no torch arithmetic on the step path."""
        model = self.model
        n = batch[model.image_latents_key if model.image_latents_key in batch else model.image_key].shape[0]
        mb = min(self.microbatch_size, n)
        starts = list(range(0, n, mb))
        total = torch.empty(1, device=model.dit.flat_buffers()["p"].device)       # a fresh scalar per step: callers keep them
        hip.check(hip.lib().md_fill_zero(total.data_ptr(), 4, torch.cuda.current_stream().cuda_stream), "md_fill_zero")
        for i, s in enumerate(starts):
            part = {k: (v[s:s + mb] if torch.is_tensor(v) and v.shape[0] == n else v) for k, v in batch.items()}
            self.sync.active = (i == len(starts) - 1)
            if self.sync.active:
                self.sync.begin_backward(single_microbatch=(len(starts) == 1))
            w = min(mb, n - s) / n
            model.train_microbatch(part, grad_scale=w, loss_accum=total, accum_weight=w)
        self.sync.active = False
        self.sync.end_backward()
        if self.measure_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                  # behind the last backward kernel on the compute stream
        slots = self.sync.finish()
        if self.measure_comm:
            e1.record()                                  # behind the waits on the collectives / side-stream norms
            self._comm_events.append((e0, e1))
        fac = self.schedule.factor(self.batches_seen) if self.schedule is not None else 1.0
        if self.measure_comm:
            o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            o0.record()
        if self.sharded:
            self.opt.step_sharded(self.sync, lr=self.opt.lr * fac, max_norm=self.clip_norm, grad_scale=1.0 / self.world,
                                  norm_slots=slots, chunk_of=self.shard_chunk_of)
            self.stale_foreign_chunks = self.world > 1
            self.model.dit.shadow_is_authoritative = self.stale_foreign_chunks
        else:
            self.opt.step(lr=self.opt.lr * fac, max_norm=self.clip_norm, grad_scale=1.0 / self.world,
                          g_bf16=self.sync.gbf, norm_partials=slots * hip.SUMSQ_PARTIALS)
        if self.measure_comm:
            o1.record()                                  # norm finish + AdamW (+ the launch of the weight all-gathers)
            self._opt_events.append((o0, o1))
        self.batches_seen += 1
        return total.reshape(())

    def exposed_comm_ms(self, last: int = 0) -> Optional[float]:
        """Mean time per step the compute stream spent waiting for the gradient exchange after the last backward kernel was
        enqueued (needs measure_comm; synchronises).  `last` > 0: only the most recent steps (skip warm-up)."""
        if not self._comm_events:
            return None
        torch.cuda.synchronize()
        ev = self._comm_events[-last:] if last > 0 else self._comm_events
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def consolidate(self) -> None:
        """Sharded optimiser only: all-gather the fp32 masters, both moments and the EMA so that every rank holds the whole,
        current state (checkpoints, evaluation on EMA weights, state_dict()).  A collective: every rank must call it.  Not on the
        step path (3 x 4.66 GB for XL/2, at checkpoint / evaluation intervals)."""
        if not self.sharded or not self.stale_foreign_chunks:
            return
        s = self.sync
        s.wait_gather()
        f = self.model.dit.flat_buffers()
        bufs = [f["p"], self.opt.m, self.opt.v] + ([self.opt.ema] if self.opt.ema is not None else [])
        for t in bufs:
            for key, lo, hi, chunk, olo in s.plan:
                mine = t[lo + s.rank * chunk: lo + (s.rank + 1) * chunk].clone()
                w = s._all_gather(t[lo:hi], mine)
                if w is not None:
                    w.wait()
        self.stale_foreign_chunks = False
        self.model.dit.shadow_is_authoritative = False

    def optimizer_ms(self, last: int = 0) -> Optional[float]:
        """Mean compute-stream time per step of the gradient-norm finish + AdamW pass (needs measure_comm; synchronises)."""
        if not self._opt_events:
            return None
        torch.cuda.synchronize()
        ev = self._opt_events[-last:] if last > 0 else self._opt_events
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def sync_replicas(self) -> None:
        """Make every rank's weights and optimiser state bit-identical to rank 0's (call once after initialisation / resume).
        Data parallelism here replicates the weights and relies on identical updates from then on; nothing else would catch a
        replica that started from a different state."""
        if self.world <= 1:
            return
        self.consolidate()          # after a sharded step rank 0's masters of foreign chunks are stale: make them whole first
        f = self.model.dit.flat_buffers()
        bufs = [f["p"], self.opt.m, self.opt.v] + ([self.opt.ema] if self.opt.ema is not None else [])
        for t in bufs:
            if self.sync.torch_bounce:
                h = t.cpu()
                dist.broadcast(h, src=0, group=self.sync.pg)
                t.copy_(h)
            else:
                dist.broadcast(t, src=0, group=self.sync.pg)
        self.model.dit.refresh_shadow(force=True)

    def replicas_in_sync(self) -> bool:
        """True when the weights of all ranks are bit-identical: an EXACT integer checksum (md_checksum_u16: sum of the bf16 bit
        patterns and an index-weighted sum, per data-parallel bucket) of the bf16 shadow every rank computes with (under the
        sharded optimiser the fp32 masters of foreign chunks are deliberately stale), compared across ranks by MIN / MAX
        all-reduces of the int64 sums.  One bf16 ulp in one weight, a sign flip or a permutation changes it (the fp32 sum of
        squares used before resolved only ~4e-3 of a bucket's 1e8 weights: ADVICE r4).  One bandwidth pass over 2.3 GB."""
        if self.world <= 1:
            return True
        self.sync.wait_gather()
        f = self.model.dit.flat_buffers()
        L, st = hip.lib(), torch.cuda.current_stream().cuda_stream
        mine = torch.zeros(len(self.sync.bucket_list), 2, device=f["s"].device, dtype=torch.int64)
        for i, (_, blo, bhi) in enumerate(self.sync.bucket_list):
            n = (bhi - blo) // 8 * 8             # buckets start on 1024-element boundaries; a ragged end (none today) is left to the next check
            hip.check(L.md_checksum_u16(f["s"].data_ptr() + 2 * blo, n, mine.data_ptr() + 16 * i, st), "md_checksum_u16")
        lo, hi = mine.clone(), mine.clone()
        if self.sync.torch_bounce:
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.sync.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.sync.pg)
        return bool(torch.equal(lo, hi))

    def throughput(self, global_batch: int, window: int = 3) -> Optional[float]:
        """Composer SpeedMonitor(window_size=3) definition: samples over the last `window` batches / wall time."""
        torch.cuda.synchronize()
        self._win.append((time.time(), self.batches_seen))
        self._win = self._win[-(window + 1):]
        if len(self._win) < 2:
            return None
        (t0, b0), (t1, b1) = self._win[0], self._win[-1]
        return (b1 - b0) * global_batch / max(t1 - t0, 1e-9)
