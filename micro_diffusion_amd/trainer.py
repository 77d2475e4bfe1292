"""Training-step runtime for the MI355X MicroDiT path: what Composer's Trainer + FSDP + GradientClipping +
torch.optim.AdamW + the LR scheduler do around `LatentDiffusion.forward` in the reference (train.py:14-123,
configs/*.yaml), re-designed for one process per GPU with RCCL over xGMI:

  * the per-rank batch (global_batch / world_size, train.py:50) is split into microbatches of
    `device_train_microbatch_size` (yaml trainer.device_train_microbatch_size); each microbatch loss is scaled by
    n_micro / n_rank_batch before backward (Composer semantics); gradients accumulate in
    the flat fp32 buffer;
  * data parallelism = gradient averaging (numerically the reference's FSDP SHARD_GRAD_OP, yaml trainer.fsdp_config):
    during the LAST microbatch's backward each finished segment (final layer, one DiT block, ...) of the flat
    gradient buffer is all-reduced asynchronously — RCCL runs on its own stream and overlaps the remaining
    backward kernels; no other collective exists on the data path;
  * after the last bucket: one fused pass computes ||g||, one fused pass clips (coef from the device-side norm, no
    host sync), applies AdamW, re-emits the bf16 shadow weights and zeroes the gradients.
"""
from __future__ import annotations

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
        self.step_count = 0
        self.ema_smoothing, self.ema_start = ema_smoothing, int(ema_start)
        self.ema = torch.zeros_like(f["p"]) if ema_smoothing is not None else None
        self.ema_live = False
        self.last_grad_scale = 1.0

    def step(self, lr: Optional[float] = None, max_norm: float = 0.0, grad_scale: float = 1.0, g_bf16: Optional[torch.Tensor] = None,
             norm_partials: int = 0) -> None:
        """`g_bf16`: take the gradients from this bf16 flat buffer (data-parallel exchange buffer) instead of the fp32
        accumulators.  `norm_partials` > 0: that many per-bucket partial sums of squares are already in self.partials."""
        f = self.dit.flat_buffers()
        L, st = hip.lib(), torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        self.last_grad_scale = grad_scale
        b1, b2 = self.betas
        ss = None
        if max_norm and max_norm > 0:
            if norm_partials <= 0:
                src = g_bf16 if g_bf16 is not None else f["g"]
                hip.check(L.md_sumsq(src.data_ptr(), 1 if g_bf16 is not None else 0, f["total"], self.partials.data_ptr(), st), "md_sumsq")
                norm_partials = hip.SUMSQ_PARTIALS
            hip.check(L.md_sumsq_finish(self.partials.data_ptr(), norm_partials, self.sumsq.data_ptr(), st), "md_sumsq_finish")
            ss = self.sumsq.data_ptr()
        ema_mode = 0
        if self.ema is not None and self.step_count > self.ema_start:
            ema_mode = 2 if self.ema_live else 1          # first EMA batch: ema <- weights (the EMA model starts as a copy)
            self.ema_live = True
        a = hip.AdamWArgs(f["p"].data_ptr(), f["g"].data_ptr(), self.m.data_ptr(), self.v.data_ptr(), f["s"].data_ptr(), ss,
                          g_bf16.data_ptr() if g_bf16 is not None else None, self.ema.data_ptr() if self.ema is not None else None,
                          f["total"], self.lr if lr is None else lr, b1, b2, self.eps, self.weight_decay,
                          1 - b1 ** self.step_count, 1 - b2 ** self.step_count, max_norm or 0.0, grad_scale,
                          self.ema_smoothing or 0.0, 1, ema_mode)
        hip.check(L.md_adamw_step(byref(a), st), "md_adamw_step")
        self.dit.mark_shadow_fresh()

    def grad_norm(self) -> torch.Tensor:
        """||g||_2 of the last step's (rank-averaged) gradient before clipping (device scalar; valid after step() with
        max_norm > 0).  The sum of squares is taken over the rank-SUMMED gradient; the 1 / world factor is applied here as in
        the optimiser kernel."""
        return self.sumsq.sqrt() * self.last_grad_scale

    def state_dict(self):
        sd = {"m": self.m, "v": self.v, "step": self.step_count}
        if self.ema is not None:
            sd["ema"], sd["ema_live"] = self.ema, self.ema_live
        return sd

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        if self.ema is not None and "ema" in sd:
            self.ema.copy_(sd["ema"])
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


class GradSync:
    """Overlapped data-parallel gradient averaging (the reference's FSDP gradient reduction, configs/*.yaml fsdp_config).

    As soon as the engine reports the backward of a segment (final layer, one DiT block, ...) as enqueued, that segment of
    the flat gradient buffer is handed to RCCL (torch.distributed, backend "nccl" = RCCL over xGMI; async: the collective
    runs on RCCL's own stream behind an event on the compute stream) while the remaining backward kernels keep the CUs busy.
    Exchange formats:
      "bf16"  the segment is cast into a bf16 staging buffer (one HIP kernel) and THAT is all-reduced: 2.33 GB instead of
              4.66 GB per step for XL/2 (FSDP's default mixed precision reduces gradients in the low precision too, SURVEY.md
              C.7); the optimiser kernel reads the reduced bf16 gradients and still zeroes the fp32 accumulators;
      "fp32"  in-place all-reduce of the fp32 accumulators (gloo / parity runs).
    The squared norm of every reduced segment is taken on a side stream right behind its collective (deterministic partial
    sums, md_sumsq), so the clip coefficient needs no extra pass after the last bucket and is bit-identical on all ranks."""

    def __init__(self, dit, process_group=None, exchange: str = "auto", single_rank_exchange: bool = False):
        """`single_rank_exchange`: run the whole exchange path (staging cast, asynchronous collective, side-stream norm) also on
        a process group of ONE rank, where the all-reduce is the identity — the only way to drive the RCCL code path on a box
        with a single GPU (tests/test_dp_gpu.py); never set in production."""
        self.dit = dit
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.enabled = self.world > 1 or (single_rank_exchange and dist.is_initialized())
        if exchange == "auto":
            exchange = "bf16" if (self.enabled and dist.get_backend(process_group) == "nccl") else "fp32"
        assert exchange in ("bf16", "fp32")
        self.exchange = exchange
        f = dit.flat_buffers()
        # segment prefix -> [start, end) in the flat buffer (table order is contiguous per module)
        import numpy as np
        self.ranges: Dict[str, List[int]] = {}
        for spec in dit._table:
            if spec.buffer:
                continue
            top = spec.name.split(".")
            key = ".".join(top[:2]) if top[0] in ("blocks", "patch_mixer") else ("final_layer" if top[0] == "final_layer" else "rest")
            o = f["offs"][spec.name]
            n = ((int(np.prod(spec.shape)) + 63) // 64) * 64
            r = self.ranges.setdefault(key, [o, o + n])
            r[0], r[1] = min(r[0], o), max(r[1], o + n)
        self.pending = []
        self.active = False
        self.buckets = 0                 # buckets handed over in the current step (= partial-sum slots in use)
        self.last_buckets = 0            # ... in the last finished step (bench.py dp block)
        self.step_bytes = self.last_bytes = 0   # bytes handed to the collective in the current / last step
        self.norm_partials: Optional[torch.Tensor] = None     # set by the Trainer: FusedAdamW.partials
        self.gbf = torch.empty(f["total"], device=f["g"].device, dtype=torch.bfloat16) if (exchange == "bf16" and self.enabled) else None
        self.side = torch.cuda.Stream(device=f["g"].device) if (self.enabled and f["g"].is_cuda) else None
        self.host_bounce = self.enabled and f["g"].is_cuda and dist.get_backend(process_group) != "nccl"

    def describe(self) -> str:
        if not self.enabled:
            return "none (single rank)"
        return f"{self.exchange} all-reduce per backward segment ({len(self.ranges) + 2} buckets), overlapped with backward"

    def _exchange(self, lo: int, hi: int) -> None:
        f = self.dit.flat_buffers()
        g = f["g"][lo:hi]
        if self.exchange == "bf16":
            buf = self.gbf[lo:hi]
            hip.check(hip.lib().md_cast_f32_bf16(g.data_ptr(), buf.data_ptr(), hi - lo, None, torch.cuda.current_stream().cuda_stream), "cast")
        else:
            buf = g
        if self.host_bounce:
            # gloo (functional runs of several ranks on one GPU, CPU tests): reduce through host memory, synchronously
            h = buf.float().cpu() if buf.dtype == torch.bfloat16 else buf.cpu()
            dist.all_reduce(h, group=self.pg)
            buf.copy_(h.to(buf.dtype))
            work = None
        else:
            work = dist.all_reduce(buf, group=self.pg, async_op=True)
        slot = self.buckets
        self.buckets += 1
        self.step_bytes += buf.numel() * buf.element_size()
        if self.norm_partials is not None and self.side is not None and slot < FusedAdamW.MAX_BUCKETS:
            if work is None:
                self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                if work is not None:
                    work.wait()          # stream-side dependency under NCCL; the norm overlaps the remaining backward
                hip.check(hip.lib().md_sumsq(buf.data_ptr(), 1 if self.exchange == "bf16" else 0, hi - lo,
                                             self.norm_partials.data_ptr() + 4 * slot * hip.SUMSQ_PARTIALS, self.side.cuda_stream), "md_sumsq")
        if work is not None:
            self.pending.append(work)

    def on_segment(self, name: str) -> None:
        if not self.active or not self.enabled:
            return
        for lo, hi in (self._rest_ranges() if name == "rest" else [tuple(self.ranges[name])]):
            self._exchange(lo, hi)

    def _rest_ranges(self):
        """'rest' = everything that is not a DiT block or the final layer; not contiguous (front end precedes the
        mixer, the mixer maps sit between mixer and backbone): reduce the gaps between the block ranges."""
        total = self.dit.flat_buffers()["total"]
        taken = sorted(v for k, v in self.ranges.items() if k != "rest")
        out, cur = [], 0
        for lo, hi in taken:
            if lo > cur:
                out.append((cur, lo))
            cur = max(cur, hi)
        if cur < total:
            out.append((cur, total))
        return out

    def finish(self) -> int:
        """Wait for every bucket; returns the number of norm partial-sum slots filled (0 = the optimiser takes the norm itself)."""
        for w in self.pending:
            w.wait()
        self.pending = []
        n = self.buckets if (self.norm_partials is not None and self.side is not None and 0 < self.buckets <= FusedAdamW.MAX_BUCKETS) else 0
        if self.side is not None and self.buckets:
            torch.cuda.current_stream().wait_stream(self.side)
        self.last_buckets, self.last_bytes = self.buckets, self.step_bytes
        self.buckets = self.step_bytes = 0
        return n


class Trainer:
    def __init__(self, model, optimizer: FusedAdamW, schedule: Optional[LRSchedule] = None, clip_norm: float = 0.0,
                 microbatch_size: int = 256, process_group=None, log: Optional[Callable[[dict], None]] = None,
                 exchange: str = "auto", single_rank_exchange: bool = False):
        self.model, self.opt, self.schedule, self.clip_norm = model, optimizer, schedule, clip_norm
        self.microbatch_size = microbatch_size
        self.sync = GradSync(model.dit, process_group, exchange=exchange, single_rank_exchange=single_rank_exchange)
        self.sync.norm_partials = optimizer.partials
        self.world = self.sync.world
        model.dit._on_segment = self.sync.on_segment
        # the Trainer runs forward -> backward strictly in turn: activations may live in the engine's fixed-address arenas
        model.dit._ensure_flat()
        model.dit.engine.use_arena = True
        self.batches_seen = 0
        self.log = log
        self._win: List[tuple] = []
        self.measure_comm = False          # bench.py: event pairs around GradSync.finish() (exposed exchange time)
        self._comm_events: List[tuple] = []

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
            w = min(mb, n - s) / n
            model.train_microbatch(part, grad_scale=w, loss_accum=total, accum_weight=w)
        self.sync.active = False
        if self.measure_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                  # behind the last backward kernel on the compute stream
        slots = self.sync.finish()
        if self.measure_comm:
            e1.record()                                  # behind the waits on the collectives / side-stream norms
            self._comm_events.append((e0, e1))
        fac = self.schedule.factor(self.batches_seen) if self.schedule is not None else 1.0
        self.opt.step(lr=self.opt.lr * fac, max_norm=self.clip_norm, grad_scale=1.0 / self.world,
                      g_bf16=self.sync.gbf, norm_partials=slots * hip.SUMSQ_PARTIALS)
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

    def sync_replicas(self) -> None:
        """Make every rank's weights and optimiser state bit-identical to rank 0's (call once after initialisation / resume).
        Data parallelism here replicates the weights and relies on identical updates from then on; nothing else would catch a
        replica that started from a different state."""
        if self.world <= 1:
            return
        f = self.model.dit.flat_buffers()
        bufs = [f["p"], self.opt.m, self.opt.v] + ([self.opt.ema] if self.opt.ema is not None else [])
        for t in bufs:
            if self.sync.host_bounce:
                h = t.cpu()
                dist.broadcast(h, src=0, group=self.sync.pg)
                t.copy_(h)
            else:
                dist.broadcast(t, src=0, group=self.sync.pg)
        self.model.dit.refresh_shadow(force=True)

    def replicas_in_sync(self) -> bool:
        """True when the fp32 master weights of all ranks have the same checksum (sum and sum of squares in fp64); cheap enough
        to run every few hundred batches."""
        if self.world <= 1:
            return True
        p = self.model.dit.flat_buffers()["p"].double()
        mine = torch.stack([p.sum(), (p * p).sum()])
        lo, hi = mine.clone(), mine.clone()
        if self.sync.host_bounce:
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
