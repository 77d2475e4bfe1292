"""DiTEngine — the hand-written forward AND backward of MicroDiT as an explicit sequence of HIP kernel launches.

No autograd below this boundary: every backward op is launched by `backward()` in reverse order of `forward()`,
reading the activations `forward()` saved in a `Tape`.  Host code is Python; every operation on device data is a
call into libmicrodit_hip.so through ctypes (micro_diffusion_amd/hip.py).  PyTorch is used for device memory
(`torch.empty`), the current HIP stream and nothing else.

Reference being replaced (paths relative to /root/reference/micro_diffusion/models):
  DiT.forward_without_cfg dit.py:455-519, DiTBlock.forward dit.py:232-239, FeedForward dit.py:88-89,
  FeedForwardECMoe.forward dit.py:126-143, AttentionBlockPromptEmbedding dit.py:53-56, SelfAttention /
  CrossAttention utils.py:116-136,178-197, T2IFinalLayer utils.py:236-240, TimestepEmbedder utils.py:283-285,
  CaptionProjection/Mlp utils.py:63-68, get_mask / mask_out_token / unmask_tokens utils.py:382-426 — and the
  autograd backward of all of them (no reference source: derived from the forward).

Numerics contract (amp_bf16 + low-precision LayerNorm): bf16 storage for activations, weights
(shadow copies of the fp32 masters) and activation gradients; fp32 accumulation in every GEMM / reduction; fp32
LayerNorm statistics, softmax, MoE routing and weight gradients (accumulated into the fp32 `.grad` buffer).
"""
from __future__ import annotations

import ctypes
import math
import os
from ctypes import byref
from typing import Dict, List, Optional

import torch

from . import hip
from .arch import BlockPlan, DiTConfig, caption_ffn_hidden, plan_blocks

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
TAIL_WS_BYTES = 256 * 256 * 256 * 4     # md_gemm_args.tail_ws: one raw 256 x 256 fp32 tile per workgroup (64 MiB of the split-K workspace)


class Tape:
    """Activations saved by one forward pass (plain attribute bag)."""
    pass


class _Arena:
    """Bump allocator over ONE device buffer.  With it the activations, gradients and scratch of a microbatch live at fixed
    addresses and the step path makes no allocator calls (no caching-allocator bookkeeping per launch, no at::native fill
    kernels; the launch sequence of a microbatch becomes capturable as a hipGraph).  The first pass through a new shape key
    runs in measuring mode — plain torch.empty, sizes recorded with the same mark / release discipline — and the buffer is
    then (re)allocated at the largest peak of all keys seen, so alternating shapes (a ragged last microbatch) do not
    re-measure.  A pass that raised is not recorded."""
    ALIGN = 256

    def __init__(self, dev):
        self.dev, self.buf, self.key = dev, None, None
        self.peaks = {}                 # shape key -> measured peak bytes
        self.top = self.peak = 0
        self.measuring = True

    def begin(self, key):
        """Start a pass of shape `key`.  The buffer is (re)allocated HERE, never at the end of a measuring pass: the tensors of
        the measuring pass (a whole activation tape: 156 GB for XL/2 at microbatch 1024) are still alive when it ends, and
        measured peak + buffer would not fit the HBM; by the next pass the caller has dropped them."""
        self.key = key
        need = self.peaks.get(key)
        if need is not None and (self.buf is None or self.buf.numel() < need + self.ALIGN):
            want = max(self.peaks.values()) + self.ALIGN
            self.buf = None             # release before re-allocating
            self.buf = torch.empty(want, device=self.dev, dtype=torch.uint8)
        self.measuring = need is None
        self.top = self.peak = 0

    def end(self, ok: bool = True):
        if self.measuring and ok:       # a pass that raised half-way did not reach the shape's peak: not recorded
            self.peaks[self.key] = self.peak

    def mark(self) -> int:
        return self.top

    def release(self, mark: int) -> None:
        self.top = mark

    def alloc(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        off = self.top
        self.top = (off + nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.peak = max(self.peak, self.top)
        if self.measuring:
            return torch.empty(*shape, device=self.dev, dtype=dtype)
        if self.top > self.buf.numel():
            raise RuntimeError("activation arena overflow: the launch sequence changed without a change of the shape key")
        return self.buf[off:off + nbytes].view(dtype).view(*shape)


def _p(t):
    return None if t is None else t.data_ptr()


class DiTEngine:
    def __init__(self, cfg: DiTConfig, params: Dict[str, torch.Tensor], shadows: Dict[str, torch.Tensor],
                 grads: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor]):
        """params: fp32 masters; shadows: bf16 copies (same names); grads: fp32 accumulators; buffers: pos_embed,
        mask_token.  All are views into flat device buffers owned by the DiT module."""
        self.cfg = cfg
        self.P, self.S, self.G, self.buf = params, shadows, grads, buffers
        self.mixer, self.backbone = plan_blocks(cfg)
        self.dev = next(iter(params.values())).device
        self.L = hip.lib()
        self.wgrad_target_blocks = 768
        self.gemm_profile = None
        self.kernel_profile = None         # bench.py: {class name: [(event0, event1, algorithmic bytes)]} for the bandwidth-bound kernels
        self.gemm_prefer = hip.GEMM_AUTO   # tests / A-B runs: a kernel to force wherever it accepts the problem (else the library's choice)
        self.attn_bwd_prefer = hip.ATTN_BWD_AUTO   # same for the attention backward (md_attn_args.bwd_split)
        # Head-major q / k (round 6): QK-LayerNorm writes q / k as [B, H, S, hd] (md_qkln_fwd_hm), attention returns dq / dk in that
        # layout and md_qkln_bwd_hm brings them back row-major.  Built, parity-tested and MEASURED NEUTRAL in the step (attention
        # -10.5 ms, QK-LayerNorm +7 ms of a 2,311 ms profile; same-box A/B 2,658 vs 2,659 images/s at microbatch 1024, -0.6 % at 256:
        # profiles/r6_head_major_qk.txt) -- v / o / dO / dv stay in packed rows unless the GEMM epilogues re-lay them.  Off by default
        # (it costs 2/3 of a qkv buffer per block of tape); MD_QK_HEAD_MAJOR=1 or this attribute turns it on.
        self.qk_head_major = os.environ.get("MD_QK_HEAD_MAJOR", "0") == "1"
        self.gemm_log = None               # tests: list that receives (variant actually requested, M, N, K, batch) per launch
        self.before_segment = None         # data parallelism: callable(bucket key) run before the first kernel that reads the bf16
        #                                    weights of a bucket ("rest", block names, "final_layer"): waits for their all-gather
        self.keep_last_tape = False        # tests: keep the most recent forward's tape in self.last_tape (per-block activations)
        self.last_tape = None
        self.route_override = None         # tests: {block name: top-k token indices [B, E, k]} replacing the router's choice
        self.ws = torch.empty(128 << 20, device=self.dev, dtype=F32)  # 512 MiB split-K workspace
        # Fixed-address activation memory (opt-in: the caller must run forward -> backward strictly in turn, as the Trainer
        # does; two forwards in flight would share the tape arena).
        self.use_arena = False
        self._tape_arena, self._scratch_arena = _Arena(self.dev), _Arena(self.dev)
        self._arena = None          # arena the next empty() / zeros() comes from (None = torch allocator)
        self._record = False        # current forward keeps per-block tapes
        self._chosen = ctypes.c_int32(-1)   # md_gemm_args.chosen_variant lands here
        self._ptr_cache = {}
        self.ksplit_min_items = 192  # work items a split-K factor must reach (A/B: 128 = the round-2 rule)
        self.splitk_force_pp = False  # A/B: force pp256 for every split-K launch it accepts, whatever the tile count
        self.short_k_accum = True   # A/B: False = split-K slices also for short contractions into large outputs
        self.group_wgrad = True     # the weight gradients of a block that contract over the same tokens as grouped launches (A/B: False)
        self._wgroup = None
        self.single_slice_ws = True  # launches with >= 192 tiles of their own: one fp32 slice through the workspace on pp256 (A/B: False = accumulate epilogue)
        self.group_dycond = True    # ONE launch for the caption-token gradients of all cross-attention kv projections of a group (A/B: False)
        self.group_adaln = True     # one launch for the condition-vector gradients of all adaLN layers of a group (A/B: False)
        self._posb = None
        self.moe_cache_dact = True  # the MoE fc1 epilogue stores gelu'(h) instead of h; the fc2 dgrad epilogue multiplies by it (A/B: False)
        self.batch_adaln = True     # the modulation of ALL blocks from one GEMM per forward (A/B: False = one small GEMM per block)
        self._adaln = self._adaln_region()
        self.gemm_tail_mode = int(os.environ.get("MD_GEMM_TAIL", "0"))   # md_gemm_args.tail_mode: 0 = the library decides, 1 = never, 2 = always (A/B)
        # One-microbatch steps of the data-parallel Trainer (a rank of the 8-GPU run: configs/res_256_pretrain.yaml:24,111): weight
        # gradients whose fp32 accumulator lies in [g_lo, g_hi) are STORED as bf16 at gbf + (addr - g_lo) / 2 -- the exchange buffer of
        # GradSync -- by the split-K reduction (or the GEMM epilogue when nothing is split): no read-modify-write of the fp32
        # accumulator, no cast + clear pass afterwards.  (g_lo, g_hi, gbf address, {fp32 address: elements stored} of this step)
        self.wgrad_bf16 = None
        self.fuse_swiglu_bwd = os.environ.get("MD_FUSE_SWIGLU_BWD", "1") != "0"   # A/B: 0 = data gradient + md_swiglu_bwd as two launches
        self.cu_limit_fn = None     # data parallelism: callable() -> CUs the persistent GEMM may occupy right now (0 = all): the
        #                             Trainer leaves the CUs of RCCL's channels free while a collective is in flight

    def _adaln_region(self):
        """The block adaLN Linear layers (dit.py:222-225: mod_l = W_l gelu(c) + b_l, the same input for every block) as ONE GEMM:
        dit.flat_layout keeps their weights contiguous in forward order (mixer blocks, then backbone blocks) and their biases
        likewise, so [W_0; W_1; ...] is a [sum 6 d_l, D] matrix.  Returns {"N", "w", "b", "col": {block name: first column}} or None
        when the tensors are not laid out that way (a foreign flat layout): the per-block GEMMs remain."""
        names = [bp.name for bp in self.mixer] + [bp.name for bp in self.backbone]
        col, n, w0, b0 = {}, 0, None, None
        for nm in names:
            w, b = self.S.get(nm + ".adaLN_modulation.1.weight"), self.P.get(nm + ".adaLN_modulation.1.bias")
            if w is None or b is None or w.dim() != 2 or w.shape[1] != self.cfg.dim:
                return None
            if w0 is None:
                w0, b0 = w, b
            elif w.data_ptr() != w0.data_ptr() + 2 * n * self.cfg.dim or b.data_ptr() != b0.data_ptr() + 4 * n:
                return None
            col[nm] = n
            n += w.shape[0]
        if w0 is None or n % 8:
            return None
        return {"N": n, "w": w0.data_ptr(), "b": b0.data_ptr(), "col": col}

    def _adaln_all(self, gc, B):
        """mod_all [B, N_all] bf16 = gelu(c) [W_0; W_1; ...]^T + [b_0; b_1; ...]: 780 output tiles at XL/2 instead of 34 launches
        of 18-24 tiles each (96 TFLOP/s at microbatch 256, profiles/r3_gemm_shapes_mb256.txt)."""
        r = self._adaln
        out = self.empty(B, r["N"])
        D = self.cfg.dim
        self._gemm(A=gc.data_ptr(), B=r["w"], C=out.data_ptr(), bias=r["b"], M=B, N=r["N"], K=D, lda=D, ldb=D, ldc=r["N"], batch=1, ksplit=1,
                   a_kcontig=1, b_kcontig=1, mode=hip.EPI_STORE_BF16, act=0, alpha=1.0)
        return out

    # ------------------------------------------------------------------------------------------ launch helpers
    def _st(self):
        return torch.cuda.current_stream().cuda_stream

    def empty(self, *shape, dtype=BF16):
        if self._arena is not None:
            return self._arena.alloc(shape, dtype)
        return torch.empty(*shape, device=self.dev, dtype=dtype)

    def zeros(self, *shape, dtype=F32):
        t = self.empty(*shape, dtype=dtype)
        hip.check(self.L.md_fill_zero(t.data_ptr(), t.numel() * t.element_size(), self._st()), "md_fill_zero")
        return t

    def _prof(self, name, nbytes, launch):
        """Run `launch()`; with kernel_profile set, bracket it with HIP events on the launch stream (bench.py roofline_hbm leg)."""
        kp = self.kernel_profile
        if kp is None:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        kp.setdefault(name, []).append((e0, e1, float(nbytes)))

    def _qkln_fwd(self, ptr, rows, ld, off, width, rstd_ptr, nseg=1):
        """nseg = 2: the q and the k half (adjacent, `width` apart) of a packed qkv row in one launch; rstd [nseg][rows]."""
        self._prof("qk_layernorm", 4.0 * rows * width * nseg, lambda: hip.check(
            self.L.md_qkln_fwd(ptr, rows, ld, off, width, nseg, width, rstd_ptr, self.cfg.norm_eps, self._st()), "md_qkln_fwd"))

    def _qkln_bwd(self, dptr, ldd, doff, yptr, ldy, yoff, rows, width, rstd_ptr, nseg=1):
        self._prof("qk_layernorm", 6.0 * rows * width * nseg, lambda: hip.check(
            self.L.md_qkln_bwd(dptr, ldd, doff, yptr, ldy, yoff, rows, width, nseg, width, width, rstd_ptr, self._st()), "md_qkln_bwd"))

    def _qkln_fwd_hm(self, ptr, rows, ld, off, width, out, S, rstd_ptr, nseg=1):
        """As _qkln_fwd, the normalised values going head-major to out [nseg][B, H, S, hd]; the input stays as it is."""
        self._prof("qk_layernorm", 4.0 * rows * width * nseg, lambda: hip.check(
            self.L.md_qkln_fwd_hm(ptr, rows, ld, off, width, nseg, width, out.data_ptr(), rows * width, S, self.cfg.head_dim, rstd_ptr,
                                  self.cfg.norm_eps, self._st()), "md_qkln_fwd_hm"))

    def _qkln_bwd_hm(self, dy, y, dptr, ldd, doff, rows, width, S, rstd_ptr, nseg=1):
        """dy, y: head-major [nseg][B, H, S, hd]; dL/dx goes row-major to dptr (segments `width` apart, as in _qkln_bwd)."""
        self._prof("qk_layernorm", 6.0 * rows * width * nseg, lambda: hip.check(
            self.L.md_qkln_bwd_hm(dy.data_ptr(), rows * width, y.data_ptr(), rows * width, dptr, ldd, doff, width, rows, width, nseg, S,
                                  self.cfg.head_dim, rstd_ptr, self._st()), "md_qkln_bwd_hm"))

    def _attn_fwd(self, a):
        nb = 2.0 * a.hd * (2 * a.Sq + 2 * a.Skv) * a.B * a.H
        self._prof("attention", nb, lambda: hip.check(self.L.md_attn_fwd(byref(a), self._st()), "md_attn_fwd"))

    def _gemm(self, **kw):
        a = hip.GemmArgs()
        problems = kw.pop("_problems", None)      # grouped launch: the problem dicts (accounting only; the table is in kw["problems"])
        may_refuse = kw.pop("_try", False)        # True: a launch the library may refuse (NOT_ELIGIBLE, nothing launched) -> returns False
        for k, v in kw.items():
            setattr(a, k, v)
        prof = self.gemm_profile
        if prof is not None:      # per-launch HIP events on the launch stream (bench.py roofline leg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        chosen = self._chosen
        a.chosen_variant = ctypes.addressof(chosen)
        if self.cu_limit_fn is not None:
            a.cu_limit = self.cu_limit_fn()
        if self.gemm_tail_mode != 1 and a.mode in (hip.EPI_STORE_BF16, hip.EPI_RESIDUAL, hip.EPI_DACT):
            # whole rounds + split-K tail (md_gemm_args.tail_ws): bf16-output launches never use the split-K workspace themselves,
            # and everything that does is ordered behind them on this stream
            a.tail_ws, a.tail_ws_bytes, a.tail_mode = self.ws.data_ptr(), TAIL_WS_BYTES, self.gemm_tail_mode
        rc = hip.NOT_ELIGIBLE
        want = self.gemm_prefer if self.gemm_prefer != hip.GEMM_AUTO else a.variant
        if want != hip.GEMM_AUTO:
            a.variant = want
            rc = self.L.md_gemm_bf16(byref(a), self._st())      # NOT_ELIGIBLE: the forced kernel refuses this problem (nothing was launched)
            if rc == hip.NOT_ELIGIBLE:
                a.variant = hip.GEMM_AUTO
        if rc == hip.NOT_ELIGIBLE:
            rc = self.L.md_gemm_bf16(byref(a), self._st())
        if rc == hip.NOT_ELIGIBLE and may_refuse:
            return False
        hip.check(rc, "md_gemm_bf16")
        if self.gemm_log is not None:      # the kernel the library actually launched (written back through chosen_variant)
            self.gemm_log.append((chosen.value, a.M, a.N, a.K, a.batch))
        if prof is not None:
            e1.record()
            # algorithmic HBM bytes of the launch: both operands once, every output once (+ fused-epilogue operands)
            out_b = 4 if a.mode in (hip.EPI_STORE_F32, hip.EPI_ACCUM_F32, hip.EPI_ATOMIC_F32) else 2
            mn = a.M * a.N * a.batch
            byt = 2.0 * (a.M * a.K + a.N * a.K) * a.batch + mn * out_b * a.ksplit
            byt += mn * 2 * ((1 if a.C2 else 0) + (1 if a.res else 0) + (1 if a.aux else 0)) + (mn * 4 if a.mode == hip.EPI_ACCUM_F32 else 0)
            if a.mode == hip.EPI_SWIGLU_BWD:
                byt += mn * 2 * 2            # h2 beside h1, dh2 beside dh1
            fl = 2.0 * a.M * a.N * a.K * a.batch
            key = (a.M, a.N, a.K, a.batch, a.a_kcontig, a.b_kcontig, a.ksplit)
            if problems:
                mn = sum(g["M"] * g["N"] for g in problems)
                fl = 2.0 * mn * a.K
                byt = 2.0 * a.K * sum(g["M"] + g["N"] for g in problems) + mn * 4 * a.ksplit
                key = (-len(problems), mn // 1024, a.K, 1, 0, 0, a.ksplit)
            prof.append((e0, e1, fl, key, byt))
        return True

    def lin_fwd(self, x, wname, out, M, N, K, *, ldx=None, ldc=None, mode=hip.EPI_STORE_BF16, act=0, res=None,
                gate=None, ldg=0, rps=0, C2=None, ldc2=0, xoff=0, ooff=0, bias=True):
        """out[M,N] = epilogue(x[M,K] @ W[N,K]^T + b).  x / out may be offset views (element offsets)."""
        b = self.P.get(wname + ".bias") if bias else None
        self._gemm(A=x.data_ptr() + 2 * xoff, B=self.S[wname + ".weight"].data_ptr(),
                   C=out.data_ptr() + (4 if mode in (hip.EPI_STORE_F32, hip.EPI_ACCUM_F32) else 2) * ooff,
                   C2=_p(C2), bias=_p(b), res=_p(res), gate=gate, M=M, N=N, K=K, lda=ldx or K, ldb=K, ldc=ldc or N,
                   ldc2=ldc2 or N, ldr=N, ldg=ldg, rows_per_sample=rps, batch=1, ksplit=1, a_kcontig=1, b_kcontig=1,
                   mode=mode, act=act, alpha=1.0)

    def lin_dgrad(self, dy, wname, dx, M, N, K, *, lddy=None, mode=hip.EPI_STORE_BF16, act=0, aux=None, res=None,
                  dyoff=0):
        """dx[M,K] (=|+=) dy[M,N] @ W[N,K]   (contraction over N; W is the K-strided operand)."""
        self._gemm(A=dy.data_ptr() + 2 * dyoff, B=self.S[wname + ".weight"].data_ptr(), C=dx.data_ptr(), aux=_p(aux),
                   res=_p(res), M=M, N=K, K=N, lda=lddy or N, ldb=K, ldc=K, ldaux=K, ldr=K, batch=1, ksplit=1,
                   a_kcontig=1, b_kcontig=0, mode=mode, act=act, alpha=1.0)

    def _fused12(self, pre):
        """True when w1.weight and w2.weight of a SwiGLU FFN are adjacent in the flat buffers (no biases in between), so
        [w1; w2] is one [2f, d] matrix: forward, dgrad and wgrad then run as ONE GEMM each."""
        w1, w2 = self.S.get(pre + ".w1.weight"), self.S.get(pre + ".w2.weight")
        if w1 is None or w2 is None or (pre + ".w1.bias") in self.P:
            return False
        return (w2.data_ptr() == w1.data_ptr() + w1.numel() * 2 and
                self.G[pre + ".w2.weight"].data_ptr() == self.G[pre + ".w1.weight"].data_ptr() + w1.numel() * 4)

    def _ksplit(self, out_rows, out_cols, contraction, batch=1):
        """Split-K factor for GEMMs whose output is too small to fill the chip (weight gradients, skinny dgrads).

        Preferred: a factor the persistent 256 x 256 kernel (pp256) accepts — the contraction per split a multiple of 128 —
        chosen by a small cost model of that kernel (profiles/r2_wgrad_splitk_pp256.txt): the tiles x splits work items
        are dealt out to the 256 CUs, a workgroup needs ~1.9 us per 64-deep k-tile plus ~5 us to write a 256 x 256 fp32
        slice, and the slice reduction moves (ks + 2) x the output once at ~4 TB/s.  Otherwise (ragged contraction,
        tiny outputs) the round-1 rule for the 2-stage kernels: ~3 workgroups of 128 x 128 per CU, >= 512 k per split."""
        ws_cap = max(1, self.ws.numel() // (out_rows * out_cols * batch))
        t256 = ((out_rows + 255) // 256) * ((out_cols + 255) // 256) * batch
        if getattr(self, "short_k_accum", True) and contraction <= 512 and t256 >= 64:
            return 1        # a short contraction into a large output (the adaLN weight gradients: 256 tokens into 6144 x 1024) is
            #                 pure output traffic: accumulate in place (one read + one write) instead of slices + a reduction pass
        if contraction % 128 == 0 and out_cols % 8 == 0:
            units = contraction // 128
            out_us = out_rows * out_cols * batch * 4 / 4e6
            best, best_cost = None, None
            # ks = 1 is a candidate too: a launch with enough tiles of its own (the 8-expert weight gradients) writes ONE fp32
            # slice to the workspace and the reduction pass adds it to the accumulator -- 3 passes over the output instead of the
            # 2 slices + 4 passes the smallest split costs, on the persistent kernel instead of the 2-stage accumulate epilogue
            for ks in range(1, min(64, ws_cap, units) + 1):
                if units % ks or t256 * ks < getattr(self, "ksplit_min_items", 192):   # md_gemm_bf16's AUTO rule sends
                    continue                                                           # < 192 tiles to the 2-stage kernels
                per_wg = -(-t256 * ks // 256)
                cost = per_wg * (contraction / ks / 64 * 1.9 + 5.0) + (ks + 2) * out_us
                if best_cost is None or cost < best_cost:
                    best, best_cost = ks, cost
            if best is not None:
                return best if best > 1 else -1          # -1: one slice through the workspace (pp256), not the accumulate epilogue
        tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128) * batch
        ks = max(1, self.wgrad_target_blocks // tiles)
        ks = min(ks, max(1, contraction // 512))
        return max(1, min(ks, ws_cap))

    def gemm_f32_acc(self, *, out_ptr, M, N, K, ldo, batch=1, sOut=0, accumulate=True, **operands):
        """fp32 C (+)= A B^T with automatic split-K: partial products go to a workspace as dense fp32 slices and are
        summed by md_splitk_reduce (deterministic; float atomics measured 4-10x slower on this shape class)."""
        ks = self._ksplit(M, N, K, batch)
        via_ws = ks != 1
        if ks == -1:                      # enough tiles without splitting: one slice through the workspace on pp256
            ks, via_ws = 1, self.single_slice_ws
        bf_ptr = self._wgrad_bf16_target(out_ptr, M * N * batch if sOut in (0, M * N) and ldo == N else None) if accumulate else None
        if bf_ptr is not None and not via_ws and ldo % 8:
            via_ws = True                 # a bf16 row pitch the GEMM epilogue cannot store (16-byte pieces): one slice + the reduction
        if not via_ws:
            if bf_ptr is not None:
                self._gemm(C=bf_ptr, M=M, N=N, K=K, ldc=ldo, sC=sOut, batch=batch, ksplit=1, mode=hip.EPI_STORE_BF16, act=0, alpha=1.0,
                           **operands)
                return
            self._gemm(C=out_ptr, M=M, N=N, K=K, ldc=ldo, sC=sOut, batch=batch, ksplit=1,
                       mode=hip.EPI_ACCUM_F32 if accumulate else hip.EPI_STORE_F32, act=0, alpha=1.0, **operands)
            return
        if self.splitk_force_pp and self.gemm_prefer == hip.GEMM_AUTO:
            operands = dict(operands, variant=hip.GEMM_PP256)
        self._gemm(C=self.ws.data_ptr(), M=M, N=N, K=K, ldc=N, sC=ks * M * N, sSplit=M * N, batch=batch, ksplit=ks,
                   mode=hip.EPI_STORE_F32, act=0, alpha=1.0, **operands)
        if bf_ptr is not None:
            self._prof("splitk_reduce", M * N * batch * (4.0 * ks + 2), lambda: hip.check(
                self.L.md_splitk_reduce(self.ws.data_ptr(), bf_ptr, M, N, ldo, sOut, ks, batch, 2, self._st()), "md_splitk_reduce"))
            return
        self._prof("splitk_reduce", 4.0 * M * N * batch * (ks + (2 if accumulate else 1)), lambda: hip.check(
            self.L.md_splitk_reduce(self.ws.data_ptr(), out_ptr, M, N, ldo, sOut, ks, batch, 1 if accumulate else 0, self._st()),
            "md_splitk_reduce"))

    def _wgrad_bf16_target(self, out_ptr, numel):
        """Address in the bf16 exchange buffer a weight gradient of this step is stored at, or None (not a one-microbatch step, the
        output is not a gradient accumulator, or the tensor is not dense).  A second gradient for the same tensor within one backward
        would have to be ADDED: no layer of the model does that; it raises instead of silently dropping the first."""
        t = getattr(self, "wgrad_bf16", None)      # (host-logic tests build the engine without __init__)
        if t is None or numel is None or not (t["g_lo"] <= out_ptr < t["g_hi"]):
            return None
        if out_ptr in t["written"]:
            raise RuntimeError("two weight gradients for one tensor in a one-microbatch step: the bf16 store path cannot accumulate")
        t["written"][out_ptr] = numel          # fp32 address -> elements stored from there (may span adjacent tensors: [w1; w2])
        return t["gbf"] + (out_ptr - t["g_lo"]) // 2

    def lin_wgrad(self, dy, x, wname, M, N, K, *, lddy=None, ldx=None, dyoff=0, xoff=0, bias_from=None, defer=False):
        """grad W[N,K] += dy[M,N]^T @ x[M,K]  (both operands K-strided; split-K over the token dimension);
        grad b[N] += column sums of dy.
        defer: with a weight-gradient group open (_wgrad_begin), only record the problem -- the whole group runs as ONE grouped
        launch at _wgrad_flush().  The caller guarantees that dy and x are not modified before the flush."""
        gb = self.G.get(wname + ".bias")
        if gb is not None:
            src = dy if bias_from is None else bias_from
            hip.check(self.L.md_colsum(src.data_ptr() + (0 if bias_from is not None else 2 * dyoff),
                                       1 if src.dtype == F32 else 0, lddy or N, gb.data_ptr(), M, N, self._st()), "colsum")
        grp = self._wgroup
        if defer and grp is not None and M % 128 == 0 and K % 8 == 0 and (not grp or grp[0]["tokens"] == M) and len(grp) < hip.GEMM_MAX_PROBLEMS:
            grp.append(dict(tokens=M, A=dy.data_ptr() + 2 * dyoff, lda=lddy or N, B=x.data_ptr() + 2 * xoff, ldb=ldx or K, M=N, N=K,
                            out=self.G[wname + ".weight"].data_ptr(), keep=(dy, x)))
            return
        self.gemm_f32_acc(out_ptr=self.G[wname + ".weight"].data_ptr(), M=N, N=K, K=M, ldo=K,
                          A=dy.data_ptr() + 2 * dyoff, B=x.data_ptr() + 2 * xoff, lda=lddy or N, ldb=ldx or K,
                          a_kcontig=0, b_kcontig=0)

    def _wgrad_begin(self):
        self._wgroup = [] if self.group_wgrad else None

    def _wgrad_flush(self):
        """Run the recorded weight gradients (same token count) as ONE grouped pp256 launch: their output tiles share the 256
        workgroups, so the split over the tokens only has to fill what ALL of them leave empty (ks = 2..8 instead of 8..16 each),
        the fp32 slices mirror the layout of the gradient tensors, and one flat reduction per contiguous run finishes them."""
        grp, self._wgroup = self._wgroup, ([] if self._wgroup is not None else None)
        if not grp:
            return
        if len(grp) == 1:
            g = grp[0]
            self.gemm_f32_acc(out_ptr=g["out"], M=g["M"], N=g["N"], K=g["tokens"], ldo=g["N"], A=g["A"], B=g["B"], lda=g["lda"],
                              ldb=g["ldb"], a_kcontig=0, b_kcontig=0)
            return
        grp.sort(key=lambda g: g["out"])
        base = grp[0]["out"]
        span = max(g["out"] + 4 * g["M"] * g["N"] for g in grp) - base
        span_el = span // 4
        tokens = grp[0]["tokens"]
        units = tokens // 128
        t256 = sum(((g["M"] + 255) // 256) * ((g["N"] + 255) // 256) for g in grp)
        out_us = sum(g["M"] * g["N"] for g in grp) * 4 / 4e6
        ws_cap = self.ws.numel() // span_el
        best, best_cost = None, None
        for ks in range(1, min(64, ws_cap, units) + 1):
            if units % ks:
                continue
            per_wg = -(-t256 * ks // 256)
            cost = per_wg * (tokens / ks / 64 * 1.9 + 5.0) + (ks + 2) * out_us
            if t256 * ks < 192:
                cost *= 192.0 / (t256 * ks)            # an under-filled chip: the round costs the same with fewer tiles done
            if best_cost is None or cost < best_cost:
                best, best_cost = ks, cost
        if best is None:                                # does not fit the workspace: one by one
            for g in grp:
                self.gemm_f32_acc(out_ptr=g["out"], M=g["M"], N=g["N"], K=g["tokens"], ldo=g["N"], A=g["A"], B=g["B"], lda=g["lda"],
                                  ldb=g["ldb"], a_kcontig=0, b_kcontig=0)
            return
        ks = best
        probs = (hip.GemmProblem * len(grp))()
        for i, g in enumerate(grp):
            probs[i] = hip.GemmProblem(g["A"], g["B"], g["lda"], g["ldb"], g["M"], g["N"], (g["out"] - base) // 4)
        g0 = grp[0]
        self._gemm(A=g0["A"], B=g0["B"], C=self.ws.data_ptr(), M=g0["M"], N=g0["N"], K=tokens, lda=g0["lda"], ldb=g0["ldb"], ldc=g0["N"],
                   sSplit=span_el, batch=1, ksplit=ks, a_kcontig=0, b_kcontig=0, mode=hip.EPI_STORE_F32, act=0, alpha=1.0,
                   problems=ctypes.addressof(probs), n_problems=len(grp), _problems=grp)
        # contiguous runs of gradient tensors -> one flat reduction each
        runs, cur = [], [grp[0]["out"], grp[0]["out"] + 4 * grp[0]["M"] * grp[0]["N"]]
        for g in grp[1:]:
            if g["out"] == cur[1]:
                cur[1] = g["out"] + 4 * g["M"] * g["N"]
            else:
                runs.append(cur)
                cur = [g["out"], g["out"] + 4 * g["M"] * g["N"]]
        runs.append(cur)
        bf = [self._wgrad_bf16_target(g["out"], g["M"] * g["N"]) for g in grp]
        to_bf16 = all(b is not None for b in bf)
        assert to_bf16 or not any(b is not None for b in bf)        # a group lies in the gradient buffer as a whole, or not at all
        for lo, hi in runs:
            n = (hi - lo) // 4
            if to_bf16:
                t = self.wgrad_bf16
                dst = t["gbf"] + (lo - t["g_lo"]) // 2
                self._prof("splitk_reduce", n * (4.0 * ks + 2), lambda lo=lo, n=n, dst=dst: hip.check(
                    self.L.md_splitk_reduce_flat(self.ws.data_ptr() + (lo - base), dst, n, span_el, ks, 2, self._st()), "md_splitk_reduce_flat"))
                continue
            self._prof("splitk_reduce", 4.0 * n * (ks + 2), lambda lo=lo, n=n: hip.check(
                self.L.md_splitk_reduce_flat(self.ws.data_ptr() + (lo - base), lo, n, span_el, ks, 1, self._st()), "md_splitk_reduce_flat"))

    def ln_args(self, x, wname, out, rows, C, *, shift=None, scale=None, ldmod=0, rps=0, mean=None, rstd=None, act=0,
                pos=None, pos_rows=0):
        return hip.LnArgs(x.data_ptr(), _p(self.P[wname + ".weight"]) if wname else None, shift, scale, _p(pos), _p(out),
                          _p(mean), _p(rstd), rows, C, C, C, ldmod, rps, pos_rows, self.cfg.norm_eps, act)

    def ln_fwd(self, a):
        self._prof("layernorm", 4.0 * a.rows * a.C, lambda: hip.check(self.L.md_ln_fwd(byref(a), self._st()), "md_ln_fwd"))

    @staticmethod
    def _rows_per_block(rows, rps, min_blocks=2048):
        """Rows of one sample per workgroup for the column-reducing backward kernels: aim at >= `min_blocks` workgroups
        (4 waves each) without dropping below 4 rows per workgroup (one per wave).  md_ln_bwd runs 4 workgroups per CU and ends each
        with an LDS reduction + 2 C atomics: 1024 workgroups (one full round) with twice the rows each beat 2048 -- 16,384 x 1024:
        35.5 -> 30.9 us, 65,536 x 1024: 100.3 -> 97.8, 65,536 x 768: 69.3 -> 68.1 (scripts/bench_norm.py --rpb-sweep)."""
        rpb = 64
        while rpb > 4 and (rows + rpb - 1) // rpb < min_blocks:
            rpb //= 2
        return int(max(1, min(rpb, rps)))

    def ln_bwd(self, a, dz, dx, *, accumulate, wname=None, dscale=None, dshift=None, ldg=0):
        """LayerNorm(+modulate) backward.  dscale / dshift: raw pointers into the (zeroed) fp32 adaLN gradient buffer for
        modulated norms; plain norms get a zeroed [samples, C] scratch for the per-sample sums the weight grad is
        finished from."""
        rps = a.rows_per_sample if a.rows_per_sample > 0 else a.rows
        rpb = self._rows_per_block(a.rows, rps, 1024)
        is_out = 1 if dscale is not None else 0
        if dscale is None and wname is not None:
            scratch = self.zeros(a.rows // rps, a.C)
            dscale, ldg = scratch.data_ptr(), a.C
        b = hip.LnBwdArgs(dz.data_ptr(), _p(dx), dscale, dshift, _p(self.G[wname + ".weight"]) if wname else None,
                          a.C, a.C, ldg, rpb, 1 if accumulate else 0, is_out)
        self._prof("layernorm", (8.0 if accumulate else 6.0) * a.rows * a.C,
                   lambda: hip.check(self.L.md_ln_bwd(byref(a), byref(b), self._st()), "md_ln_bwd"))

    def attn_args(self, q, k, v, o, lse, B, H, Sq, Skv, ldq, ldk, ldv, hid, *, do=None, dq=None, dk=None, dv=None,
                  delta=None, lddq=0, lddk=0, lddv=0, hm_qk=False):
        """hm_qk: q, k (and dq, dk) are head-major [B, H, S, hd] buffers (ldq / ldk / lddq / lddk are ignored)."""
        hd = self.cfg.head_dim
        a = hip.AttnArgs(q, k, v, _p(o), _p(lse), _p(do), dq, dk, dv, _p(delta), B, H, Sq, Skv, ldq, ldk, ldv, hid,
                         Sq * ldq, Skv * ldk, Skv * ldv, Sq * hid, lddq, lddk, lddv, hid, Sq * lddq, Skv * lddk,
                         Skv * lddv, Sq * hid, 1.0 / math.sqrt(hd), hd, 0)
        if hm_qk:
            a.ldq = a.ldk = a.lddq = a.lddk = hd
            a.sq = a.sdq = H * Sq * hd
            a.sk = a.sdk = H * Skv * hd
            a.hsq = a.hsdq = Sq * hd
            a.hsk = a.hsdk = Skv * hd
        return a

    def _attn_bwd(self, a):
        self._prof("attention", 2.0 * a.hd * (4 * a.Sq + 4 * a.Skv) * a.B * a.H, lambda: self._attn_bwd_launch(a))

    def _attn_bwd_launch(self, a):
        rc = -1
        if self.attn_bwd_prefer != hip.ATTN_BWD_AUTO:
            a.bwd_split = self.attn_bwd_prefer
            rc = self.L.md_attn_bwd(byref(a), self._st())     # -1: the forced kernel does not cover this problem (nothing launched)
            a.bwd_split = hip.ATTN_BWD_AUTO
        if rc == -1:
            rc = self.L.md_attn_bwd(byref(a), self._st())
        hip.check(rc, "md_attn_bwd")

    # ------------------------------------------------------------------------------------------ attention layers
    def _self_attn_fwd(self, pre, xin, B, S, dim, hid, heads, t):
        """xin [B*S, dim] -> o [B*S, hid]; saves qkv (post-LN), rstd, lse on tape t."""
        M, L, st = B * S, self.L, self._st()
        qkv = self.empty(M, 3 * hid)
        self.lin_fwd(xin, pre + ".qkv", qkv, M, 3 * hid, dim)
        rq = self.empty(2, M, dtype=F32)
        o = self.empty(M, hid)
        lse = self.empty(B, heads, S, dtype=F32)
        if self.qk_head_major:
            t.qk = self.empty(2, M, hid)          # normalised q, k: [2][B, H, S, hd]; qkv keeps the raw q / k (dead) and v
            self._qkln_fwd_hm(qkv.data_ptr(), M, 3 * hid, 0, hid, t.qk, S, rq.data_ptr(), nseg=2)
            a = self.attn_args(t.qk.data_ptr(), t.qk.data_ptr() + 2 * M * hid, qkv.data_ptr() + 4 * hid, o, lse, B, heads, S, S,
                               0, 0, 3 * hid, hid, hm_qk=True)
        else:
            t.qk = None
            self._qkln_fwd(qkv.data_ptr(), M, 3 * hid, 0, hid, rq.data_ptr(), nseg=2)
            a = self.attn_args(qkv.data_ptr(), qkv.data_ptr() + 2 * hid, qkv.data_ptr() + 4 * hid, o, lse, B, heads, S, S,
                               3 * hid, 3 * hid, 3 * hid, hid)
        self._attn_fwd(a)
        t.qkv, t.rq, t.o, t.lse = qkv, rq, o, lse
        return o

    def _self_attn_bwd(self, pre, xin, do, B, S, dim, hid, heads, t, defer=False):
        """do [M, hid] -> returns d_xin [M, dim]; accumulates the qkv weight grad."""
        M, L, st = B * S, self.L, self._st()
        dqkv = self.empty(M, 3 * hid)
        delta = self.empty(B, heads, S, dtype=F32)
        qkv = t.qkv
        if t.qk is not None:
            dqk = self.empty(2, M, hid)           # dL/d(normalised q, k), head-major like t.qk
            a = self.attn_args(t.qk.data_ptr(), t.qk.data_ptr() + 2 * M * hid, qkv.data_ptr() + 4 * hid, t.o, t.lse, B, heads, S, S,
                               0, 0, 3 * hid, hid, do=do, dq=dqk.data_ptr(), dk=dqk.data_ptr() + 2 * M * hid,
                               dv=dqkv.data_ptr() + 4 * hid, delta=delta, lddv=3 * hid, hm_qk=True)
            self._attn_bwd(a)
            self._qkln_bwd_hm(dqk, t.qk, dqkv.data_ptr(), 3 * hid, 0, M, hid, S, t.rq.data_ptr(), nseg=2)
        else:
            a = self.attn_args(qkv.data_ptr(), qkv.data_ptr() + 2 * hid, qkv.data_ptr() + 4 * hid, t.o, t.lse, B, heads, S, S,
                               3 * hid, 3 * hid, 3 * hid, hid, do=do, dq=dqkv.data_ptr(), dk=dqkv.data_ptr() + 2 * hid,
                               dv=dqkv.data_ptr() + 4 * hid, delta=delta, lddq=3 * hid, lddk=3 * hid, lddv=3 * hid)
            self._attn_bwd(a)
            self._qkln_bwd(dqkv.data_ptr(), 3 * hid, 0, qkv.data_ptr(), 3 * hid, 0, M, hid, t.rq.data_ptr(), nseg=2)
        self.lin_wgrad(dqkv, xin, pre + ".qkv", M, 3 * hid, dim, defer=defer)
        dxin = self.empty(M, dim)
        self.lin_dgrad(dqkv, pre + ".qkv", dxin, M, 3 * hid, dim)
        return dxin

    # ------------------------------------------------------------------------------------------ DiT block
    def _block_fwd(self, bp: BlockPlan, x, ycond, B, S, Lc, gc, mod_all=None):
        L, st, cfg = self.L, self._st(), self.cfg
        d, h, hx, f = bp.dim, bp.attn_hidden, bp.xattn_hidden, bp.ffn_hidden
        M, Mc, D = B * S, B * Lc, cfg.dim
        n = bp.name
        t = Tape()
        t.x = x
        if mod_all is not None:              # this block's columns of the batched modulation GEMM (leading dimension N_all)
            mod, ldm = mod_all, mod_all.shape[1]
            mp = mod_all.data_ptr() + 2 * self._adaln["col"][n]
        else:
            mod, ldm = self.empty(B, 6 * d), 6 * d
            self.lin_fwd(gc, n + ".adaLN_modulation.1", mod, B, 6 * d, D)
            mp = mod.data_ptr()
        t.mod, t.mp, t.ldm = mod, mp, ldm
        # -- self attention: x1 = x + gate_msa * proj(attn(modulate(LN1(x))))
        t.xm1 = self.empty(M, d)
        t.st1 = self.empty(2, M, dtype=F32)
        a1 = self.ln_args(x, n + ".norm1", t.xm1, M, d, shift=mp, scale=mp + 2 * d, ldmod=ldm, rps=S, mean=t.st1[0], rstd=t.st1[1])
        self.ln_fwd(a1)
        t.sa = Tape()
        o = self._self_attn_fwd(n + ".attn", t.xm1, B, S, d, h, bp.heads, t.sa)
        t.br1 = self.empty(M, d)
        x1 = self.empty(M, d)
        self.lin_fwd(o, n + ".attn.proj", x1, M, d, h, mode=hip.EPI_RESIDUAL, res=x, gate=mp + 2 * 2 * d, ldg=ldm, rps=S,
                     C2=t.br1)
        t.x1 = x1
        # -- cross attention: x2 = x1 + proj(attn(q(LN2(x1)), kv(y)))
        t.xn2 = self.empty(M, d)
        t.st2 = self.empty(2, M, dtype=F32)
        self.ln_fwd(self.ln_args(x1, n + ".norm2", t.xn2, M, d, mean=t.st2[0], rstd=t.st2[1]))
        t.o2 = self.empty(M, hx)
        # head-major: the raw q is dead once md_qkln_fwd_hm has read it -- it is staged in the buffer attention then writes o into
        t.q2 = t.o2 if self.qk_head_major else self.empty(M, hx)
        self.lin_fwd(t.xn2, n + ".cross_attn.q_linear", t.q2, M, hx, d)
        t.kv = self.empty(Mc, 2 * hx)
        self.lin_fwd(ycond, n + ".cross_attn.kv_linear", t.kv, Mc, 2 * hx, d)
        t.rq2 = self.empty(M, dtype=F32)
        t.rk2 = self.empty(Mc, dtype=F32)
        t.lse2 = self.empty(B, bp.xheads, S, dtype=F32)
        if self.qk_head_major:
            t.q2n, t.k2n = self.empty(M, hx), self.empty(Mc, hx)        # normalised q [B, H, S, hd], k [B, H, Lc, hd]
            self._qkln_fwd_hm(t.q2.data_ptr(), M, hx, 0, hx, t.q2n, S, t.rq2.data_ptr())
            self._qkln_fwd_hm(t.kv.data_ptr(), Mc, 2 * hx, 0, hx, t.k2n, Lc, t.rk2.data_ptr())
            ax = self.attn_args(t.q2n.data_ptr(), t.k2n.data_ptr(), t.kv.data_ptr() + 2 * hx, t.o2, t.lse2, B, bp.xheads, S, Lc,
                                0, 0, 2 * hx, hx, hm_qk=True)
        else:
            t.q2n = t.k2n = None
            self._qkln_fwd(t.q2.data_ptr(), M, hx, 0, hx, t.rq2.data_ptr())
            self._qkln_fwd(t.kv.data_ptr(), Mc, 2 * hx, 0, hx, t.rk2.data_ptr())
            ax = self.attn_args(t.q2.data_ptr(), t.kv.data_ptr(), t.kv.data_ptr() + 2 * hx, t.o2, t.lse2, B, bp.xheads, S, Lc,
                                hx, 2 * hx, 2 * hx, hx)
        self._attn_fwd(ax)
        x2 = self.empty(M, d)
        self.lin_fwd(t.o2, n + ".cross_attn.proj", x2, M, d, hx, mode=hip.EPI_RESIDUAL, res=x1)
        t.x2 = x2
        # -- feed-forward: x3 = x2 + gate_mlp * mlp(modulate(LN3(x2)))
        t.xm3 = self.empty(M, d)
        t.st3 = self.empty(2, M, dtype=F32)
        self.ln_fwd(self.ln_args(x2, n + ".norm3", t.xm3, M, d, shift=mp + 2 * 3 * d, scale=mp + 2 * 4 * d, ldmod=ldm,
                                 rps=S, mean=t.st3[0], rstd=t.st3[1]))
        x3 = self.empty(M, d)
        t.br3 = self.empty(M, d)
        gate_mlp = mp + 2 * 5 * d
        if not bp.moe:
            t.h12 = self.empty(M, 2 * f)
            if self._fused12(n + ".mlp"):
                self.lin_fwd(t.xm3, n + ".mlp.w1", t.h12, M, 2 * f, d)          # [w1; w2] as one [2f, d] weight
            else:
                self.lin_fwd(t.xm3, n + ".mlp.w1", t.h12, M, f, d, ldc=2 * f)
                self.lin_fwd(t.xm3, n + ".mlp.w2", t.h12, M, f, d, ldc=2 * f, ooff=f)
            t.a = self.empty(M, f)
            self._prof("swiglu", 6.0 * M * f, lambda: hip.check(L.md_swiglu_fwd(t.h12.data_ptr(), 2 * f, t.a.data_ptr(), f, M, f, st), "swiglu"))
            self.lin_fwd(t.a, n + ".mlp.w3", x3, M, d, f, mode=hip.EPI_RESIDUAL, res=x2, gate=gate_mlp, ldg=ldm, rps=S,
                         C2=t.br3)
        else:
            E = cfg.num_experts
            k = int(cfg.expert_capacity * S / E)
            ldl = 8 if E <= 8 else 16
            Bk = B * k
            t.k, t.ldl = k, ldl
            logits = self.empty(M, ldl, dtype=F32)
            self.lin_fwd(t.xm3, n + ".mlp.gate", logits, M, E, d, ldc=ldl, mode=hip.EPI_STORE_F32)
            t.probs = self.empty(M, ldl, dtype=F32)
            t.rowidx = self.empty(E, Bk, dtype=I32)
            t.gval = self.empty(E, Bk, dtype=F32)
            t.slot = self.empty(M, E, dtype=I32)
            hip.check(L.md_moe_route(logits.data_ptr(), t.probs.data_ptr(), ldl, B, S, E, k, t.rowidx.data_ptr(),
                                     t.gval.data_ptr(), t.slot.data_ptr(), st), "moe_route")
            if self.route_override is not None and n in self.route_override:
                self._override_routing(t, self.route_override[n], B, S, E, k)
            t.xin = self.empty(E, Bk, d)
            hip.check(L.md_gather_rows(t.xm3.data_ptr(), d, t.rowidx.data_ptr(), t.xin.data_ptr(), d, E * Bk, d, st), "gather")
            t.hpre = self.empty(E, Bk, f)         # the pre-activation -- or, with moe_cache_dact, gelu'(pre-activation): nothing but
            t.hact = self.empty(E, Bk, f)         # the backward's activation derivative ever reads it (md_gemm_args.dact_cached)
            t.hpre_is_dact = 1 if self.moe_cache_dact else 0
            w1, w2 = self.S[n + ".mlp.w1"], self.S[n + ".mlp.w2"]       # [E, d, f], [E, f, d]
            self._gemm(A=t.xin.data_ptr(), B=w1.data_ptr(), C=t.hact.data_ptr(), C2=t.hpre.data_ptr(), M=Bk, N=f, K=d, lda=d,
                       ldb=f, ldc=f, ldc2=f, sA=Bk * d, sB=d * f, sC=Bk * f, sC2=Bk * f, batch=E, ksplit=1, a_kcontig=1,
                       b_kcontig=0, mode=hip.EPI_STORE_BF16, act=hip.ACT_GELU_ERF, alpha=1.0, dact_cached=t.hpre_is_dact)
            t.h2 = self.empty(E, Bk, d)
            self._gemm(A=t.hact.data_ptr(), B=w2.data_ptr(), C=t.h2.data_ptr(), M=Bk, N=d, K=f, lda=f, ldb=d, ldc=d,
                       sA=Bk * f, sB=f * d, sC=Bk * d, batch=E, ksplit=1, a_kcontig=1, b_kcontig=0, mode=hip.EPI_STORE_BF16,
                       act=0, alpha=1.0)
            hip.check(L.md_moe_combine(t.h2.data_ptr(), t.gval.data_ptr(), t.slot.data_ptr(), x2.data_ptr(), gate_mlp, ldm,
                                       t.br3.data_ptr(), x3.data_ptr(), B, S, E, k, d, st), "moe_combine")
        return x3, t

    def _override_routing(self, t, m_idx, B, S, E, k):
        """Parity-test hook (never on the product path): replace the expert-choice selection of one layer by given token
        indices [B, E, k] (the fp32 oracle's top-k), keeping this run's own probabilities as gate values, so that what remains
        of a difference to the oracle is arithmetic, not a slot ranked differently by bf16 gate logits."""
        m_idx = m_idx.to(self.dev).long()
        rows = (m_idx + (torch.arange(B, device=self.dev) * S).view(B, 1, 1)).permute(1, 0, 2).reshape(E, B * k)
        t.rowidx.copy_(rows.to(I32))
        ecol = torch.arange(E, device=self.dev).view(E, 1).expand(E, B * k)
        t.gval.copy_(t.probs[rows, ecol])
        slot = torch.full((B, S, E), -1, device=self.dev, dtype=I32)
        for e in range(E):
            slot[:, :, e].scatter_(1, m_idx[:, e, :], torch.arange(k, device=self.dev, dtype=I32).view(1, k).expand(B, k))
        t.slot.copy_(slot.view(B * S, E))

    def _block_bwd(self, bp: BlockPlan, t: Tape, dx, ycond, dycond_f32, B, S, Lc, gc, dgc_f32, group=None, kvgroup=None):
        """dx: grad w.r.t. the block output [M, d] (bf16), updated IN PLACE to the grad w.r.t. the block input."""
        L, st, cfg = self.L, self._st(), self.cfg
        d, h, hx, f = bp.dim, bp.attn_hidden, bp.xattn_hidden, bp.ffn_hidden
        M, Mc, D = B * S, B * Lc, cfg.dim
        n = bp.name
        mp, ldm = t.mp, t.ldm
        self._wgrad_begin()
        dmod = self.zeros(B, 6 * d)          # fp32 grads of (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp)
        dmp = dmod.data_ptr()
        rpb = self._rows_per_block(M, S)
        # ---------------- feed-forward branch
        dbr3 = self.empty(M, d)
        self._prof("gate_bwd", 6.0 * M * d, lambda: hip.check(L.md_gate_bwd(dx.data_ptr(), t.br3.data_ptr(), mp + 2 * 5 * d, ldm, dbr3.data_ptr(),
                                                                          dmp + 4 * 5 * d, 6 * d, M, d, S, rpb, st), "gate_bwd"))
        dxm3 = self.empty(M, d)
        if not bp.moe:
            self.lin_wgrad(dbr3, t.a, n + ".mlp.w3", M, d, f, defer=True)
            dh12 = self.empty(M, 2 * f)
            # da = dbr3 @ W3 with the SwiGLU backward in the GEMM epilogue (MD_EPI_SWIGLU_BWD: h12 read, dh12 written by the launch
            # that forms da -- da never exists in memory, md_swiglu_bwd's pass over 10 bytes per element is gone); the library
            # refuses what its 4-wave kernel does not cover (ragged tiles, a CU hold): then the two separate launches
            fused = self.fuse_swiglu_bwd and self._gemm(
                A=dbr3.data_ptr(), B=self.S[n + ".mlp.w3.weight"].data_ptr(), C=dh12.data_ptr(), aux=t.h12.data_ptr(), M=M, N=f, K=d, lda=d,
                ldb=f, ldc=2 * f, ldaux=2 * f, batch=1, ksplit=1, a_kcontig=1, b_kcontig=0, mode=hip.EPI_SWIGLU_BWD, act=0, alpha=1.0, _try=True)
            if not fused:
                da = self.empty(M, f)
                self.lin_dgrad(dbr3, n + ".mlp.w3", da, M, d, f)
                self._prof("swiglu", 10.0 * M * f, lambda: hip.check(L.md_swiglu_bwd(da.data_ptr(), f, t.h12.data_ptr(), 2 * f, dh12.data_ptr(), 2 * f, M, f, st), "swiglu_bwd"))
            if self._fused12(n + ".mlp"):
                self.lin_wgrad(dh12, t.xm3, n + ".mlp.w1", M, 2 * f, d, defer=True)
                self.lin_dgrad(dh12, n + ".mlp.w1", dxm3, M, 2 * f, d)
            else:
                self.lin_wgrad(dh12, t.xm3, n + ".mlp.w1", M, f, d, lddy=2 * f, defer=True)
                self.lin_wgrad(dh12, t.xm3, n + ".mlp.w2", M, f, d, lddy=2 * f, dyoff=f, defer=True)
                self.lin_dgrad(dh12, n + ".mlp.w1", dxm3, M, f, d, lddy=2 * f)
                self.lin_dgrad(dh12, n + ".mlp.w2", dxm3, M, f, d, lddy=2 * f, dyoff=f, mode=hip.EPI_RESIDUAL, res=dxm3)
        else:
            E, k, ldl = cfg.num_experts, t.k, t.ldl
            Bk = B * k
            w1, w2 = self.S[n + ".mlp.w1"], self.S[n + ".mlp.w2"]
            g1, g2 = self.G[n + ".mlp.w1"], self.G[n + ".mlp.w2"]
            dh2 = self.empty(E, Bk, d)
            dgval = self.empty(E, Bk, dtype=F32)
            hip.check(L.md_moe_combine_bwd(dbr3.data_ptr(), t.h2.data_ptr(), t.rowidx.data_ptr(), t.gval.data_ptr(),
                                           dh2.data_ptr(), dgval.data_ptr(), E * Bk, d, st), "combine_bwd")
            # dW2[e][f, d] += hact[e]^T dh2[e]
            self.gemm_f32_acc(out_ptr=g2.data_ptr(), M=f, N=d, K=Bk, ldo=d, batch=E, sOut=f * d, A=t.hact.data_ptr(),
                              B=dh2.data_ptr(), lda=f, ldb=d, sA=Bk * f, sB=Bk * d, a_kcontig=0, b_kcontig=0)
            # dhpre = (dh2 @ W2[e]^T) * gelu'(hpre)
            dhpre = self.empty(E, Bk, f)
            self._gemm(A=dh2.data_ptr(), B=w2.data_ptr(), C=dhpre.data_ptr(), aux=t.hpre.data_ptr(), M=Bk, N=f, K=d, lda=d,
                       ldb=d, ldc=f, ldaux=f, sA=Bk * d, sB=f * d, sC=Bk * f, sAux=Bk * f, batch=E, ksplit=1, a_kcontig=1,
                       b_kcontig=1, mode=hip.EPI_DACT, act=hip.ACT_GELU_ERF, alpha=1.0, dact_cached=t.hpre_is_dact)
            # dW1[e][d, f] += xin[e]^T dhpre[e]
            self.gemm_f32_acc(out_ptr=g1.data_ptr(), M=d, N=f, K=Bk, ldo=f, batch=E, sOut=d * f, A=t.xin.data_ptr(),
                              B=dhpre.data_ptr(), lda=d, ldb=f, sA=Bk * d, sB=Bk * f, a_kcontig=0, b_kcontig=0)
            # dxin = dhpre @ W1[e]^T
            dxin = self.empty(E, Bk, d)
            self._gemm(A=dhpre.data_ptr(), B=w1.data_ptr(), C=dxin.data_ptr(), M=Bk, N=d, K=f, lda=f, ldb=f, ldc=d, sA=Bk * f,
                       sB=d * f, sC=Bk * d, batch=E, ksplit=1, a_kcontig=1, b_kcontig=1, mode=hip.EPI_STORE_BF16, act=0, alpha=1.0)
            dlog = self.empty(M, ldl)
            hip.check(L.md_moe_dispatch_bwd(dxin.data_ptr(), t.slot.data_ptr(), dxm3.data_ptr(), t.probs.data_ptr(), ldl,
                                            dgval.data_ptr(), dlog.data_ptr(), ldl, B, S, E, k, d, st), "dispatch_bwd")
            # gate: dWg[E, d] += dlog^T xm3 ; dxm3 += dlog @ Wg
            self.lin_wgrad(dlog, t.xm3, n + ".mlp.gate", M, E, d, lddy=ldl)
            self.lin_dgrad(dlog, n + ".mlp.gate", dxm3, M, E, d, lddy=ldl, mode=hip.EPI_RESIDUAL, res=dxm3)
        a3 = self.ln_args(t.x2, n + ".norm3", None, M, d, scale=mp + 2 * 4 * d, ldmod=ldm, rps=S, mean=t.st3[0], rstd=t.st3[1])
        self.ln_bwd(a3, dxm3, dx, accumulate=True, wname=n + ".norm3", dscale=dmp + 4 * 4 * d, dshift=dmp + 4 * 3 * d, ldg=6 * d)
        # ---------------- cross-attention branch (un-gated, un-modulated)
        self.lin_wgrad(dx, t.o2, n + ".cross_attn.proj", M, d, hx, defer=True)     # dx stays as it is until the flush below
        do2 = self.empty(M, hx)
        self.lin_dgrad(dx, n + ".cross_attn.proj", do2, M, d, hx)
        dq2 = self.empty(M, hx)
        dkv = self.empty(Mc, 2 * hx) if kvgroup is None else kvgroup["buf"][len(kvgroup["w"])]
        delta = self.empty(B, bp.xheads, S, dtype=F32)
        if t.q2n is not None:
            dq2n, dk2n = self.empty(M, hx), self.empty(Mc, hx)          # head-major like t.q2n / t.k2n
            ax = self.attn_args(t.q2n.data_ptr(), t.k2n.data_ptr(), t.kv.data_ptr() + 2 * hx, t.o2, t.lse2, B, bp.xheads, S, Lc, 0,
                                0, 2 * hx, hx, do=do2, dq=dq2n.data_ptr(), dk=dk2n.data_ptr(), dv=dkv.data_ptr() + 2 * hx,
                                delta=delta, lddv=2 * hx, hm_qk=True)
            self._attn_bwd(ax)
            self._qkln_bwd_hm(dq2n, t.q2n, dq2.data_ptr(), hx, 0, M, hx, S, t.rq2.data_ptr())
            self._qkln_bwd_hm(dk2n, t.k2n, dkv.data_ptr(), 2 * hx, 0, Mc, hx, Lc, t.rk2.data_ptr())
        else:
            ax = self.attn_args(t.q2.data_ptr(), t.kv.data_ptr(), t.kv.data_ptr() + 2 * hx, t.o2, t.lse2, B, bp.xheads, S, Lc, hx,
                                2 * hx, 2 * hx, hx, do=do2, dq=dq2.data_ptr(), dk=dkv.data_ptr(), dv=dkv.data_ptr() + 2 * hx,
                                delta=delta, lddq=hx, lddk=2 * hx, lddv=2 * hx)
            self._attn_bwd(ax)
            self._qkln_bwd(dq2.data_ptr(), hx, 0, t.q2.data_ptr(), hx, 0, M, hx, t.rq2.data_ptr())
            self._qkln_bwd(dkv.data_ptr(), 2 * hx, 0, t.kv.data_ptr(), 2 * hx, 0, Mc, hx, t.rk2.data_ptr())
        self.lin_wgrad(dq2, t.xn2, n + ".cross_attn.q_linear", M, hx, d, defer=True)
        self.lin_wgrad(dkv, ycond, n + ".cross_attn.kv_linear", Mc, 2 * hx, d)
        # d(ycond) accumulates in fp32 over all blocks that attend to these caption tokens: per block (split-K slices + a
        # reduction pass into the [B*L, d] fp32 buffer, 28 times), or -- kvgroup -- deferred: dkv of every block is kept and ONE
        # launch contracts the concatenation [dkv_0 | dkv_1 | ...] with [Wkv_0; Wkv_1; ...] (_dycond_grouped)
        if kvgroup is None:
            self.gemm_f32_acc(out_ptr=dycond_f32.data_ptr(), M=Mc, N=d, K=2 * hx, ldo=d, A=dkv.data_ptr(),
                              B=self.S[n + ".cross_attn.kv_linear.weight"].data_ptr(), lda=2 * hx, ldb=d, a_kcontig=1, b_kcontig=0)
        else:
            kvgroup["w"].append(self.S[n + ".cross_attn.kv_linear.weight"].data_ptr())
        dxn2 = self.empty(M, d)
        self.lin_dgrad(dq2, n + ".cross_attn.q_linear", dxn2, M, hx, d)
        self._wgrad_flush()                  # w3, [w1; w2], cross_attn.proj, q_linear -- before dx (cross_attn.proj's dy) is updated
        a2 = self.ln_args(t.x1, n + ".norm2", None, M, d, mean=t.st2[0], rstd=t.st2[1], rps=S)
        self.ln_bwd(a2, dxn2, dx, accumulate=True, wname=n + ".norm2")
        # ---------------- self-attention branch
        dbr1 = self.empty(M, d)
        self._prof("gate_bwd", 6.0 * M * d, lambda: hip.check(L.md_gate_bwd(dx.data_ptr(), t.br1.data_ptr(), mp + 2 * 2 * d, ldm, dbr1.data_ptr(),
                                                                          dmp + 4 * 2 * d, 6 * d, M, d, S, rpb, st), "gate_bwd"))
        self.lin_wgrad(dbr1, t.sa.o, n + ".attn.proj", M, d, h, defer=True)
        do = self.empty(M, h)
        self.lin_dgrad(dbr1, n + ".attn.proj", do, M, d, h)
        dxm1 = self._self_attn_bwd(n + ".attn", t.xm1, do, B, S, d, h, bp.heads, t.sa, defer=True)
        self._wgrad_flush()                  # attn.proj, attn.qkv
        a1 = self.ln_args(t.x, n + ".norm1", None, M, d, scale=mp + 2 * d, ldmod=ldm, rps=S, mean=t.st1[0], rstd=t.st1[1])
        self.ln_bwd(a1, dxm1, dx, accumulate=True, wname=n + ".norm1", dscale=dmp + 4 * d, dshift=dmp, ldg=6 * d)
        self._wgroup = None
        # ---------------- adaLN linear: mod = W gelu(c) + b
        self._adaln_bwd(n + ".adaLN_modulation.1", dmod, B, 6 * d, gc, dgc_f32, group)

    def _adaln_bwd(self, wname, dmod_f32, B, N, gc, dgc_f32, group=None):
        """Backward of mod = W gelu(c) + b for one layer.  The weight / bias gradients are taken at once; the gradient w.r.t.
        the condition vector, dgc += dmod @ W (a [B, N] x [N, D] product per layer, too small to fill the chip: 62 launches of
        ~60 us per microbatch of 256 at ~50 TFLOP/s, profiles/r2_kernel_stats_bench_mb256.txt), is deferred when `group` is
        given: the bf16 dmod goes to the group's slot and ALL layers of the group are contracted by one launch over operand
        lists at the end of the backward (_adaln_dgrad_grouped)."""
        D = self.cfg.dim
        if group is not None:
            i = len(group["w"])
            dmod = group["buf"][i]
        else:
            dmod = self.empty(B, N)
        hip.check(self.L.md_cast_f32_bf16(dmod_f32.data_ptr(), dmod.data_ptr(), B * N, None, self._st()), "cast")
        self.lin_wgrad(dmod, gc, wname, B, N, D, bias_from=dmod_f32)
        if group is not None:
            group["w"].append(self.S[wname + ".weight"].data_ptr())
            return
        self.gemm_f32_acc(out_ptr=dgc_f32.data_ptr(), M=B, N=D, K=N, ldo=D, A=dmod.data_ptr(),
                          B=self.S[wname + ".weight"].data_ptr(), lda=N, ldb=D, a_kcontig=1, b_kcontig=0)

    def _dycond_grouped(self, group, Mc, d, K, out_f32):
        """out[Mc, d] += sum_l dkv_l[Mc, K] @ Wkv_l[K, d] over the G blocks of a group: ONE pp256 launch whose items walk the
        operand lists segment by segment (md_gemm_args.list_segments) and keep the sum in their accumulators; split over
        `ks` groups of segments only as far as whole rounds of the 256 CUs need it (a handful of fp32 slices instead of
        2 x 28)."""
        G = len(group["w"])
        if G == 0:
            return
        t256 = ((Mc + 255) // 256) * ((d + 255) // 256)
        if K % 128 or d % 8 or Mc * d > self.ws.numel():
            for i in range(G):
                self.gemm_f32_acc(out_ptr=out_f32.data_ptr(), M=Mc, N=d, K=K, ldo=d, A=group["buf"][i].data_ptr(), B=group["w"][i],
                                  lda=K, ldb=d, a_kcontig=1, b_kcontig=0)
            return
        out_us = Mc * d * 4 / 4e6
        best, best_cost = 1, None
        for ks in range(1, G + 1):
            if G % ks or ks * Mc * d > self.ws.numel():
                continue
            rounds = -(-t256 * ks // 256)
            cost = rounds * ((G // ks) * K / 64 * 1.9 + 5.0) + (ks + 2) * out_us
            if best_cost is None or cost < best_cost:
                best, best_cost = ks, cost
        ks, S = best, G // best
        base = group["buf"].data_ptr()
        al = self._ptr_list([base + 2 * i * Mc * K for i in range(G)])
        bl = self._ptr_list(group["w"])
        self._gemm(A=base, B=group["w"][0], A_list=al.data_ptr(), B_list=bl.data_ptr(), list_segments=S, C=self.ws.data_ptr(), M=Mc, N=d,
                   K=K * G, lda=K, ldb=d, ldc=d, sC=ks * Mc * d, sSplit=Mc * d, batch=1, ksplit=ks, a_kcontig=1, b_kcontig=0,
                   mode=hip.EPI_STORE_F32, act=0, alpha=1.0)
        hip.check(self.L.md_splitk_reduce(self.ws.data_ptr(), out_f32.data_ptr(), Mc, d, d, 0, ks, 1, 1, self._st()), "md_splitk_reduce")

    def _ptr_list(self, ptrs):
        """Device array of device pointers (md_gemm_args.A_list / B_list); cached by value: with the fixed-address arenas the
        same lists recur every microbatch, so the step path uploads nothing."""
        key = tuple(ptrs)
        t = self._ptr_cache.get(key)
        if t is None:
            if len(self._ptr_cache) > 256:
                self._ptr_cache.clear()
            t = torch.tensor(list(key), dtype=torch.int64).to(self.dev)
            self._ptr_cache[key] = t
        return t

    def _adaln_dgrad_grouped(self, group, B, N, dgc_f32):
        """dgc[B, D] += sum_l dmod_l[B, N] @ W_l[N, D] over the G layers of a group as ONE pp256 launch: every (layer, k-part)
        is an item with its own operand pair (md_gemm_args.A_list / B_list) writing an fp32 slice, then one md_splitk_reduce."""
        D, G = self.cfg.dim, len(group["w"])
        if G == 0:
            return
        t256 = ((B + 255) // 256) * ((D + 255) // 256)
        parts = 1
        for c in (1, 2, 3, 4, 6, 8, 12):                      # k-parts per layer: enough items to fill the chip
            if N % c == 0 and (N // c) % 128 == 0 and (N // c) >= 128:
                parts = c
                if t256 * G * c >= 192:
                    break
        kspan = N // parts
        ks = G * parts
        if ks * B * D > self.ws.numel() or (N // parts) % 128:
            for i in range(G):                                 # does not fit the workspace / the kernel: layer by layer
                self.gemm_f32_acc(out_ptr=dgc_f32.data_ptr(), M=B, N=D, K=N, ldo=D, A=group["buf"][i].data_ptr(), B=group["w"][i],
                                  lda=N, ldb=D, a_kcontig=1, b_kcontig=0)
            return
        base = group["buf"].data_ptr()
        al = self._ptr_list([base + 2 * (i * B * N + c * kspan) for i in range(G) for c in range(parts)])
        bl = self._ptr_list([w + 2 * (c * kspan * D) for w in group["w"] for c in range(parts)])
        self._gemm(A=base, B=group["w"][0], A_list=al.data_ptr(), B_list=bl.data_ptr(), C=self.ws.data_ptr(), M=B, N=D, K=kspan * ks,
                   lda=N, ldb=D, ldc=D, sC=ks * B * D, sSplit=B * D, batch=1, ksplit=ks, a_kcontig=1, b_kcontig=0,
                   mode=hip.EPI_STORE_F32, act=0, alpha=1.0)
        hip.check(self.L.md_splitk_reduce(self.ws.data_ptr(), dgc_f32.data_ptr(), B, D, D, 0, ks, 1, 1, self._st()), "md_splitk_reduce")

    # ------------------------------------------------------------------------------------------ small MLPs (Mlp with norm)
    def _mlp_norm_fwd(self, pre, xin, rows, cin, rps, res=None):
        """fc2(LN(gelu(fc1(x)))) (utils.py:63-68); optional residual add on the output.  Returns (out, tape)."""
        D = self.cfg.dim
        t = Tape()
        t.xin = xin
        t.h = self.empty(rows, D)
        self.lin_fwd(xin, pre + ".fc1", t.h, rows, D, cin)
        t.hn = self.empty(rows, D)
        t.st = self.empty(2, rows, dtype=F32)
        self.ln_fwd(self.ln_args(t.h, pre + ".norm", t.hn, rows, D, mean=t.st[0], rstd=t.st[1], act=hip.ACT_GELU_TANH, rps=rps))
        out = self.empty(rows, D)
        if res is None:
            self.lin_fwd(t.hn, pre + ".fc2", out, rows, D, D)
        else:
            self.lin_fwd(t.hn, pre + ".fc2", out, rows, D, D, mode=hip.EPI_RESIDUAL, res=res)
        return out, t

    def _mlp_norm_bwd(self, pre, t, dout, rows, cin, rps, need_dx):
        D = self.cfg.dim
        self.lin_wgrad(dout, t.hn, pre + ".fc2", rows, D, D)
        dhn = self.empty(rows, D)
        self.lin_dgrad(dout, pre + ".fc2", dhn, rows, D, D)
        dh = self.empty(rows, D)
        a = self.ln_args(t.h, pre + ".norm", None, rows, D, mean=t.st[0], rstd=t.st[1], act=hip.ACT_GELU_TANH, rps=rps)
        self.ln_bwd(a, dhn, dh, accumulate=False, wname=pre + ".norm")
        self.lin_wgrad(dh, t.xin, pre + ".fc1", rows, D, cin)
        if not need_dx:
            return None
        dx = self.empty(rows, cin)
        self.lin_dgrad(dh, pre + ".fc1", dx, rows, D, cin)
        return dx

    # ------------------------------------------------------------------------------------------ whole model
    def forward(self, x_img: torch.Tensor, t_in: torch.Tensor, y: torch.Tensor, *, mask_ratio: float = 0.0,
                mask_noise: Optional[torch.Tensor] = None, in_scale: Optional[torch.Tensor] = None,
                y_rowscale: Optional[torch.Tensor] = None, record_tape: bool = False, arena: bool = False) -> Tape:
        """x_img f32 [B,C,H,W] (multiplied by in_scale[b] if given), t_in f32 [B], y f16|f32 [B,(1,)L,Dc]
        (rows multiplied by y_rowscale[b] if given).  Returns the tape; tape.out_tok is the network output for the
        kept tokens, bf16 [B*Tk, p*p*C].
        record_tape: a backward() will follow — keep every block's activations (inference passes drop them as they go).
        arena: additionally place the tape in the fixed-address arena (needs `use_arena`; the caller promises forward ->
        backward strictly in turn).  Both are explicit arguments: torch disables grad mode inside autograd.Function.forward,
        so probing torch.is_grad_enabled() here would never see a training pass."""
        cfg, L, st = self.cfg, self.L, self._st()
        B = x_img.shape[0]
        C, H, W, p = cfg.in_channels, x_img.shape[-2], x_img.shape[-1], cfg.patch_size
        T, D, Dm = (H // p) * (W // p), cfg.dim, cfg.patch_mixer_dim
        assert T == cfg.tokens, "input resolution does not match the model's position table"
        Lc, Dc = y.shape[-2], y.shape[-1]
        assert x_img.dtype == F32 and x_img.is_contiguous() and y.is_contiguous() and y.dtype in (torch.float16, F32)
        grad_pass = self.use_arena and arena and record_tape
        self._record = record_tape or self.keep_last_tape
        if grad_pass:
            self._tape_arena.begin((B, H, W, Lc, Dc, float(mask_ratio), str(y.dtype)))
            self._arena = self._tape_arena
        ok = False
        try:
            tp = self._forward(x_img, t_in, y, mask_ratio, mask_noise, in_scale, y_rowscale)
            ok = True
            return tp
        finally:
            if grad_pass:
                self._tape_arena.end(ok)
            self._arena = None

    def _forward(self, x_img, t_in, y, mask_ratio, mask_noise, in_scale, y_rowscale) -> Tape:
        cfg, L, st = self.cfg, self.L, self._st()
        B = x_img.shape[0]
        C, H, W, p = cfg.in_channels, x_img.shape[-2], x_img.shape[-1], cfg.patch_size
        T, D, Dm = (H // p) * (W // p), cfg.dim, cfg.patch_mixer_dim
        Lc, Dc = y.shape[-2], y.shape[-1]
        seg = self.before_segment if self.before_segment is not None else (lambda key: None)
        seg("rest")
        tp = Tape()
        tp.arena = self._arena is not None
        tp.B, tp.T, tp.Lc, tp.H, tp.W = B, T, Lc, H, W
        # ---- patch embedding (+ pos), dit.py:479
        tp.patches = self.empty(B * T, cfg.patch_vec)
        hip.check(L.md_patchify(x_img.data_ptr(), _p(in_scale), tp.patches.data_ptr(), B, C, H, W, p, st), "patchify")
        tok = self.empty(B * T, D)
        self._gemm(A=tp.patches.data_ptr(), B=self.S["x_embedder.proj.weight"].data_ptr(), C=tok.data_ptr(),
                   bias=self.P["x_embedder.proj.bias"].data_ptr(), M=B * T, N=D, K=cfg.patch_vec, lda=cfg.patch_vec,
                   ldb=cfg.patch_vec, ldc=D, batch=1, ksplit=1, a_kcontig=1, b_kcontig=1, mode=hip.EPI_STORE_BF16, act=0, alpha=1.0)
        tp.tok = tok
        # ---- timestep embedding, dit.py:480
        tp.tfreq = self.empty(B, 512)
        t_f = t_in if (t_in.dtype == F32 and t_in.numel() == B and t_in.is_contiguous()) else t_in.to(F32).expand(B).contiguous()
        hip.check(L.md_timestep_embed(t_f.data_ptr(), tp.tfreq.data_ptr(), B, 512, st), "timestep_embed")
        tp.t_pre = self.empty(B, D)
        tp.t_h = self.empty(B, D)
        self.lin_fwd(tp.tfreq, "t_embedder.mlp.0", tp.t_h, B, D, 512, act=hip.ACT_GELU_TANH, C2=tp.t_pre)
        temb = self.empty(B, D)
        self.lin_fwd(tp.t_h, "t_embedder.mlp.2", temb, B, D, D)
        # ---- caption projection + caption block, dit.py:482-483
        Mc = B * Lc
        tp.ycap = self.empty(Mc, Dc)
        hip.check(L.md_cast_rows_bf16(y.data_ptr(), 0 if y.dtype == torch.float16 else 1, tp.ycap.data_ptr(), Mc, Dc,
                                      _p(y_rowscale), Lc, st), "cast_rows")
        y0, tp.yproj = self._mlp_norm_fwd("y_embedder.y_proj", tp.ycap, Mc, Dc, Lc)
        tp.y0 = y0
        cb = Tape()
        cb.xn1 = self.empty(Mc, D)
        cb.st1 = self.empty(2, Mc, dtype=F32)
        self.ln_fwd(self.ln_args(y0, "y_emb_preprocess.norm1", cb.xn1, Mc, D, mean=cb.st1[0], rstd=cb.st1[1], rps=Lc))
        cb.sa = Tape()
        heads_c = D // cfg.head_dim
        o = self._self_attn_fwd("y_emb_preprocess.attn", cb.xn1, B, Lc, D, D, heads_c, cb.sa)
        y1 = self.empty(Mc, D)
        self.lin_fwd(o, "y_emb_preprocess.attn.proj", y1, Mc, D, D, mode=hip.EPI_RESIDUAL, res=y0)
        cb.y1 = y1
        cb.xn2 = self.empty(Mc, D)
        cb.st2 = self.empty(2, Mc, dtype=F32)
        self.ln_fwd(self.ln_args(y1, "y_emb_preprocess.norm2", cb.xn2, Mc, D, mean=cb.st2[0], rstd=cb.st2[1], rps=Lc))
        fc = caption_ffn_hidden(cfg)
        cb.h12 = self.empty(Mc, 2 * fc)
        if self._fused12("y_emb_preprocess.mlp"):
            self.lin_fwd(cb.xn2, "y_emb_preprocess.mlp.w1", cb.h12, Mc, 2 * fc, D)
        else:
            self.lin_fwd(cb.xn2, "y_emb_preprocess.mlp.w1", cb.h12, Mc, fc, D, ldc=2 * fc)
            self.lin_fwd(cb.xn2, "y_emb_preprocess.mlp.w2", cb.h12, Mc, fc, D, ldc=2 * fc, ooff=fc)
        cb.a = self.empty(Mc, fc)
        self._prof("swiglu", 6.0 * Mc * fc, lambda: hip.check(L.md_swiglu_fwd(cb.h12.data_ptr(), 2 * fc, cb.a.data_ptr(), fc, Mc, fc, st), "swiglu"))
        y2 = self.empty(Mc, D)
        self.lin_fwd(cb.a, "y_emb_preprocess.mlp.w3", y2, Mc, D, fc, mode=hip.EPI_RESIDUAL, res=y1)
        tp.cb, tp.y2 = cb, y2
        # ---- pooled caption -> condition vector c = t_emb + Mlp(mean(y)), dit.py:484-485
        tp.ymean = self.empty(B, D)
        hip.check(L.md_mean_tokens(y2.data_ptr(), tp.ymean.data_ptr(), B, Lc, D, st), "mean_tokens")
        c, tp.pool = self._mlp_norm_fwd("pooled_y_emb_process", tp.ymean, B, D, 1, res=temb)
        tp.c = c
        gc = self.empty(B, D)
        hip.check(L.md_act_fwd(c.data_ptr(), gc.data_ptr(), B * D, hip.ACT_GELU_TANH, st), "act_fwd")
        tp.gc = gc
        # ---- patch mixer, dit.py:489-493
        pos = self.buf["pos_embed"]
        if cfg.use_patch_mixer and cfg.has_maps:
            tp.xin_ln = self.empty(B * T, D)
            tp.st_xin = self.empty(2, B * T, dtype=F32)
            self.ln_fwd(self.ln_args(tok, "patch_mixer_map_xin.0", tp.xin_ln, B * T, D, mean=tp.st_xin[0], rstd=tp.st_xin[1],
                                     pos=pos, pos_rows=T, rps=T))
            x = self.empty(B * T, Dm)
            self.lin_fwd(tp.xin_ln, "patch_mixer_map_xin.1", x, B * T, Dm, D)
            tp.y_ln = self.empty(Mc, D)
            tp.st_y = self.empty(2, Mc, dtype=F32)
            self.ln_fwd(self.ln_args(y2, "patch_mixer_map_y.0", tp.y_ln, Mc, D, mean=tp.st_y[0], rstd=tp.st_y[1], rps=Lc))
            ym = self.empty(Mc, Dm)
            self.lin_fwd(tp.y_ln, "patch_mixer_map_y.1", ym, Mc, Dm, D)
        else:
            # Identity maps (patch_mixer_dim == dim, dit.py:389-392): x = tok + pos
            x = self.empty(B * T, D)
            if self._posb is None or self._posb.shape[0] != B:          # constant table: built once per batch size
                self._posb = pos.to(BF16).expand(B, T, D).contiguous()
            posb = self._posb
            hip.check(L.md_add_bf16(tok.data_ptr(), posb.data_ptr(), x.data_ptr(), B * T * D, st), "add")
            ym = y2
        tp.ym = ym
        # ---- the modulation vectors of every block (mixer + backbone) from one GEMM on gelu(c), dit.py:222-225
        seg("adaln.m")
        seg("adaln.b")
        mod_all = self._adaln_all(gc, B) if (self.batch_adaln and self._adaln is not None) else None
        tp.mixer = []
        for bp in self.mixer:
            seg(bp.name)
            x, bt = self._block_fwd(bp, x, ym, B, T, Lc, gc, mod_all)
            if self._record:
                tp.mixer.append(bt)
        # ---- masking, dit.py:495-504
        Wd = x.shape[1]
        if mask_ratio > 0:
            Tk = int(T * (1 - mask_ratio))
            assert mask_noise is not None and mask_noise.dtype == F32 and mask_noise.shape == (B, T)
            tp.keep_rows = self.empty(B * Tk, dtype=I32)
            tp.ids_restore = self.empty(B, T, dtype=I32)
            tp.mask = self.empty(B, T, dtype=F32)
            hip.check(L.md_get_mask(mask_noise.contiguous().data_ptr(), B, T, Tk, tp.keep_rows.data_ptr(),
                                    tp.ids_restore.data_ptr(), tp.mask.data_ptr(), st), "get_mask")
            xk = self.empty(B * Tk, Wd)
            hip.check(L.md_gather_rows(x.data_ptr(), Wd, tp.keep_rows.data_ptr(), xk.data_ptr(), Wd, B * Tk, Wd, st), "gather")
            x = xk
        else:
            Tk = T
            tp.keep_rows = tp.ids_restore = tp.mask = None
        tp.Tk = Tk
        # ---- mixer -> backbone projection (after masking), dit.py:506-508
        if cfg.use_patch_mixer and cfg.has_maps:
            tp.xout_in = x
            tp.xout_ln = self.empty(B * Tk, Dm)
            tp.st_xout = self.empty(2, B * Tk, dtype=F32)
            self.ln_fwd(self.ln_args(x, "patch_mixer_map_xout.0", tp.xout_ln, B * Tk, Dm, mean=tp.st_xout[0],
                                     rstd=tp.st_xout[1], rps=Tk))
            xb = self.empty(B * Tk, D)
            self.lin_fwd(tp.xout_ln, "patch_mixer_map_xout.1", xb, B * Tk, D, Dm)
            x = xb
        tp.blocks = []
        for bp in self.backbone:
            seg(bp.name)
            x, bt = self._block_fwd(bp, x, y2, B, Tk, Lc, gc, mod_all)
            if self._record:
                tp.blocks.append(bt)
        # ---- final layer, dit.py:513
        seg("final_layer")
        tp.xlast = x
        tp.fmod = self.empty(B, 2 * D)
        self.lin_fwd(gc, "final_layer.adaLN_modulation.1", tp.fmod, B, 2 * D, D)
        fm = tp.fmod.data_ptr()
        tp.xf = self.empty(B * Tk, D)
        tp.st_f = self.empty(2, B * Tk, dtype=F32)
        self.ln_fwd(self.ln_args(x, "final_layer.norm_final", tp.xf, B * Tk, D, shift=fm, scale=fm + 2 * D, ldmod=2 * D, rps=Tk,
                                 mean=tp.st_f[0], rstd=tp.st_f[1]))
        pv = cfg.patch_vec
        tp.out_tok = self.empty(B * Tk, pv)
        self.lin_fwd(tp.xf, "final_layer.linear", tp.out_tok, B * Tk, pv, D)
        if self.keep_last_tape:
            self.last_tape = tp
        return tp

    def sample_image(self, tp: Tape) -> torch.Tensor:
        """unmask_tokens + unpatchify (utils.py:417-426, dit.py:566-575) -> f32 [B, C, H, W]."""
        cfg = self.cfg
        img = self.empty(tp.B, cfg.in_channels, tp.H, tp.W, dtype=F32)
        mt = self.buf["mask_token"]
        hip.check(self.L.md_unpatchify(tp.out_tok.data_ptr(), _p(tp.ids_restore), tp.Tk, mt.data_ptr(), img.data_ptr(), tp.B,
                                       cfg.in_channels, tp.H, tp.W, cfg.patch_size, self._st()), "unpatchify")
        return img

    def backward(self, tp: Tape, dtok: torch.Tensor, on_segment=None) -> None:
        """dtok: bf16 [B*Tk, p*p*C] grad of the network output for the kept tokens.  Accumulates every parameter
        gradient into the fp32 grad buffers (self.G).  `on_segment(prefix)` is called as soon as every kernel that
        writes the gradients of the parameters under `prefix` has been enqueued (data-parallel bucket hand-off)."""
        use = self.use_arena and getattr(tp, "arena", False)
        if use:
            self._scratch_arena.begin((tp.B, tp.T, tp.Tk, tp.Lc, tp.H, tp.W))
            self._arena = self._scratch_arena
        ok = False
        try:
            self._backward(tp, dtok, on_segment)
            ok = True
        finally:
            if use:
                self._scratch_arena.end(ok)
            self._arena = None

    def _block_bwd_scoped(self, *a):
        """One block's backward; its temporaries are released when it returns (bump-arena mark / release)."""
        if self._arena is None:
            return self._block_bwd(*a)
        m = self._arena.mark()
        self._block_bwd(*a)
        self._arena.release(m)

    def _backward(self, tp: Tape, dtok: torch.Tensor, on_segment=None) -> None:
        cfg, L, st = self.cfg, self.L, self._st()
        seg = on_segment if on_segment is not None else (lambda name: None)
        B, T, Tk, Lc = tp.B, tp.T, tp.Tk, tp.Lc
        D, Dm, pv = cfg.dim, cfg.patch_mixer_dim, cfg.patch_vec
        Mc = B * Lc
        gc = tp.gc
        dgc = self.zeros(B, D)                      # fp32: sums the adaLN dgrads of all 6+28+1 layers
        dy2_f32 = self.zeros(Mc, D)                 # fp32: caption-token grads from the 28 backbone kv_linears + pooling
        grp_b = grp_m = None
        if self.group_adaln:                        # bf16 dmod of every layer, kept until the grouped contraction below
            if self.backbone:
                grp_b = {"buf": self.empty(len(self.backbone), B, 6 * self.backbone[0].dim), "w": []}
            if self.mixer:
                grp_m = {"buf": self.empty(len(self.mixer), B, 6 * self.mixer[0].dim), "w": []}
        kv_b = kv_m = None
        if self.group_dycond:                       # dkv of every block, kept until the concatenated contraction
            if self.backbone:
                kv_b = {"buf": self.empty(len(self.backbone), Mc, 2 * self.backbone[0].xattn_hidden), "w": []}
            if self.mixer:
                kv_m = {"buf": self.empty(len(self.mixer), Mc, 2 * self.mixer[0].xattn_hidden), "w": []}
        # ---- final layer
        self.lin_wgrad(dtok, tp.xf, "final_layer.linear", B * Tk, pv, D)
        dxf = self.empty(B * Tk, D)
        self.lin_dgrad(dtok, "final_layer.linear", dxf, B * Tk, pv, D)
        dfm = self.zeros(B, 2 * D)
        fm = tp.fmod.data_ptr()
        dx = self.empty(B * Tk, D)
        af = self.ln_args(tp.xlast, "final_layer.norm_final", None, B * Tk, D, scale=fm + 2 * D, ldmod=2 * D, rps=Tk,
                          mean=tp.st_f[0], rstd=tp.st_f[1])
        self.ln_bwd(af, dxf, dx, accumulate=False, wname="final_layer.norm_final", dscale=dfm.data_ptr() + 4 * D,
                    dshift=dfm.data_ptr(), ldg=2 * D)
        self._adaln_bwd("final_layer.adaLN_modulation.1", dfm, B, 2 * D, gc, dgc)
        seg("final_layer")
        # ---- backbone
        for bp, bt in zip(reversed(self.backbone), reversed(tp.blocks)):
            self._block_bwd_scoped(bp, bt, dx, tp.y2, dy2_f32, B, Tk, Lc, gc, dgc, grp_b, kv_b)
            seg(bp.name)
        if kv_b is not None:
            self._dycond_grouped(kv_b, Mc, self.backbone[0].dim, 2 * self.backbone[0].xattn_hidden, dy2_f32)
        # ---- mixer -> backbone projection
        if cfg.use_patch_mixer and cfg.has_maps:
            self.lin_wgrad(dx, tp.xout_ln, "patch_mixer_map_xout.1", B * Tk, D, Dm)
            dln = self.empty(B * Tk, Dm)
            self.lin_dgrad(dx, "patch_mixer_map_xout.1", dln, B * Tk, D, Dm)
            dxk = self.empty(B * Tk, Dm)
            a = self.ln_args(tp.xout_in, "patch_mixer_map_xout.0", None, B * Tk, Dm, mean=tp.st_xout[0], rstd=tp.st_xout[1], rps=Tk)
            self.ln_bwd(a, dln, dxk, accumulate=False, wname="patch_mixer_map_xout.0")
            dx = dxk
        Wd = dx.shape[1]
        # ---- un-mask: scatter kept-token grads back to all T positions (zeros elsewhere)
        if tp.keep_rows is not None:
            dfull = self.zeros(B * T, Wd, dtype=BF16)
            hip.check(L.md_scatter_rows(dx.data_ptr(), Wd, tp.keep_rows.data_ptr(), dfull.data_ptr(), Wd, B * Tk, Wd, st), "scatter")
            dx = dfull
        # ---- patch mixer
        has_maps = cfg.use_patch_mixer and cfg.has_maps
        dym_f32 = self.zeros(Mc, Dm) if has_maps else dy2_f32
        for bp, bt in zip(reversed(self.mixer), reversed(tp.mixer)):
            self._block_bwd_scoped(bp, bt, dx, tp.ym, dym_f32, B, T, Lc, gc, dgc, grp_m, kv_m)
            seg(bp.name)
        if kv_m is not None:
            self._dycond_grouped(kv_m, Mc, self.mixer[0].dim, 2 * self.mixer[0].xattn_hidden, dym_f32)
        # ---- map_xin / patch embedding
        if has_maps:
            self.lin_wgrad(dx, tp.xin_ln, "patch_mixer_map_xin.1", B * T, Dm, D)
            dln = self.empty(B * T, D)
            self.lin_dgrad(dx, "patch_mixer_map_xin.1", dln, B * T, Dm, D)
            dtok_e = self.empty(B * T, D)
            a = self.ln_args(tp.tok, "patch_mixer_map_xin.0", None, B * T, D, mean=tp.st_xin[0], rstd=tp.st_xin[1],
                             pos=self.buf["pos_embed"], pos_rows=T, rps=T)
            self.ln_bwd(a, dln, dtok_e, accumulate=False, wname="patch_mixer_map_xin.0")
        else:
            dtok_e = dx
        self.gemm_f32_acc(out_ptr=self.G["x_embedder.proj.weight"].data_ptr(), M=D, N=pv, K=B * T, ldo=pv,
                          A=dtok_e.data_ptr(), B=tp.patches.data_ptr(), lda=D, ldb=pv, a_kcontig=0, b_kcontig=0)
        hip.check(L.md_colsum(dtok_e.data_ptr(), 0, D, self.G["x_embedder.proj.bias"].data_ptr(), B * T, D, st), "colsum")
        # ---- condition vector: c = temb + pooled ; gc = gelu(c)
        if grp_b is not None:
            self._adaln_dgrad_grouped(grp_b, B, 6 * self.backbone[0].dim, dgc)
        if grp_m is not None:
            self._adaln_dgrad_grouped(grp_m, B, 6 * self.mixer[0].dim, dgc)
        dc = self.empty(B, D)
        hip.check(L.md_act_bwd(dgc.data_ptr(), tp.c.data_ptr(), dc.data_ptr(), B * D, hip.ACT_GELU_TANH, st), "act_bwd")
        # timestep embedder
        self.lin_wgrad(dc, tp.t_h, "t_embedder.mlp.2", B, D, D)
        dtpre = self.empty(B, D)
        self.lin_dgrad(dc, "t_embedder.mlp.2", dtpre, B, D, D, mode=hip.EPI_DACT, act=hip.ACT_GELU_TANH, aux=tp.t_pre)
        self.lin_wgrad(dtpre, tp.tfreq, "t_embedder.mlp.0", B, D, 512)
        # pooled-caption MLP -> mean over tokens
        dymean = self._mlp_norm_bwd("pooled_y_emb_process", tp.pool, dc, B, D, 1, need_dx=True)
        hip.check(L.md_mean_tokens_bwd(dymean.data_ptr(), dy2_f32.data_ptr(), B, Lc, D, st), "mean_bwd")
        # ---- caption tokens: total grad w.r.t. y2
        dy = self.empty(Mc, D)
        hip.check(L.md_cast_f32_bf16(dy2_f32.data_ptr(), dy.data_ptr(), Mc * D, None, st), "cast")
        if has_maps:
            dym = self.empty(Mc, Dm)
            hip.check(L.md_cast_f32_bf16(dym_f32.data_ptr(), dym.data_ptr(), Mc * Dm, None, st), "cast")
            self.lin_wgrad(dym, tp.y_ln, "patch_mixer_map_y.1", Mc, Dm, D)
            dyl = self.empty(Mc, D)
            self.lin_dgrad(dym, "patch_mixer_map_y.1", dyl, Mc, Dm, D)
            a = self.ln_args(tp.y2, "patch_mixer_map_y.0", None, Mc, D, mean=tp.st_y[0], rstd=tp.st_y[1], rps=Lc)
            self.ln_bwd(a, dyl, dy, accumulate=True, wname="patch_mixer_map_y.0")
        # ---- caption block (y2 = y1 + ffn(LN2(y1)); y1 = y0 + attn(LN1(y0)))
        cb = tp.cb
        fc = caption_ffn_hidden(cfg)
        self.lin_wgrad(dy, cb.a, "y_emb_preprocess.mlp.w3", Mc, D, fc)
        da = self.empty(Mc, fc)
        self.lin_dgrad(dy, "y_emb_preprocess.mlp.w3", da, Mc, D, fc)
        dh12 = self.empty(Mc, 2 * fc)
        self._prof("swiglu", 10.0 * Mc * fc, lambda: hip.check(L.md_swiglu_bwd(da.data_ptr(), fc, cb.h12.data_ptr(), 2 * fc, dh12.data_ptr(), 2 * fc, Mc, fc, st), "swiglu_bwd"))
        dxn2 = self.empty(Mc, D)
        if self._fused12("y_emb_preprocess.mlp"):
            self.lin_wgrad(dh12, cb.xn2, "y_emb_preprocess.mlp.w1", Mc, 2 * fc, D)
            self.lin_dgrad(dh12, "y_emb_preprocess.mlp.w1", dxn2, Mc, 2 * fc, D)
        else:
            self.lin_wgrad(dh12, cb.xn2, "y_emb_preprocess.mlp.w1", Mc, fc, D, lddy=2 * fc)
            self.lin_wgrad(dh12, cb.xn2, "y_emb_preprocess.mlp.w2", Mc, fc, D, lddy=2 * fc, dyoff=fc)
            self.lin_dgrad(dh12, "y_emb_preprocess.mlp.w1", dxn2, Mc, fc, D, lddy=2 * fc)
            self.lin_dgrad(dh12, "y_emb_preprocess.mlp.w2", dxn2, Mc, fc, D, lddy=2 * fc, dyoff=fc, mode=hip.EPI_RESIDUAL, res=dxn2)
        a = self.ln_args(cb.y1, "y_emb_preprocess.norm2", None, Mc, D, mean=cb.st2[0], rstd=cb.st2[1], rps=Lc)
        self.ln_bwd(a, dxn2, dy, accumulate=True, wname="y_emb_preprocess.norm2")
        heads_c = D // cfg.head_dim
        self.lin_wgrad(dy, cb.sa.o, "y_emb_preprocess.attn.proj", Mc, D, D)
        do = self.empty(Mc, D)
        self.lin_dgrad(dy, "y_emb_preprocess.attn.proj", do, Mc, D, D)
        dxn1 = self._self_attn_bwd("y_emb_preprocess.attn", cb.xn1, do, B, Lc, D, D, heads_c, cb.sa)
        a = self.ln_args(tp.y0, "y_emb_preprocess.norm1", None, Mc, D, mean=cb.st1[0], rstd=cb.st1[1], rps=Lc)
        self.ln_bwd(a, dxn1, dy, accumulate=True, wname="y_emb_preprocess.norm1")
        # ---- caption projection (inputs need no grad)
        self._mlp_norm_bwd("y_embedder.y_proj", tp.yproj, dy, Mc, tp.ycap.shape[1], Lc, need_dx=False)
        seg("rest")
