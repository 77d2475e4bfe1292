"""Drop-in `DiT` for the MI355X path: same constructor, `state_dict()` keys / shapes, initialisation and
`forward(x, t, y, cfg=1.0, mask_ratio=0) -> {'sample', 'mask'}` contract as the reference
(/root/reference/micro_diffusion/models/dit.py:249-575), but the module holds no compute: parameters are views
into one flat fp32 buffer (plus a bf16 shadow and an fp32 gradient buffer of the same layout) and the
forward/backward are executed by `DiTEngine` (HIP kernels).  Autograd sees the whole network as ONE function.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .arch import DiTConfig, ParamSpec, adaln_order, bucket_key, is_block_adaln, param_table, plan_blocks, sincos_table
from .engine import DiTEngine

_ALIGN = 64  # elements; keeps every tensor 256-byte aligned in the fp32 buffers and 128-byte in the bf16 shadow


_BUCKET_ALIGN = 1024  # elements; every data-parallel bucket starts (and therefore ends) on a multiple of it, so a bucket splits
#                       into equal, 128-byte aligned rank chunks for every world size that divides 16 (2, 4, 8, 16: reduce-scatter /
#                       all-gather in place).  Other world sizes (3, 5, 6, 7) keep the all-reduce exchange (trainer.GradSync "auto")


def flat_layout(table):
    """Offsets (in elements) of every parameter inside the flat buffers: matrix-shaped tensors in table order (= the reference's
    registration order, so the tensors of a block are contiguous and [w1; w2] of a SwiGLU stay adjacent; the modulation weights of
    all blocks first, contiguous, as the two buckets "adaln.m" / "adaln.b"), each padded to _ALIGN, a new data-parallel bucket (arch.bucket_key)
    starting on a multiple of _BUCKET_ALIGN; then all one-dimensional tensors (the "small" bucket, the block adaLN biases first)
    in one region at the end."""
    offs, total, prev = {}, 0, None
    for small in (False, True):
        # the block adaLN tensors lead their region, contiguous and in forward order (arch.is_block_adaln): ONE [sum 6 d_l, D] matrix
        # and ONE bias vector for the batched modulation GEMM
        lead = sorted((s for s in table if not s.buffer and is_block_adaln(s.name)), key=lambda s: adaln_order(s.name))
        rest = [s for s in table if not s.buffer and not is_block_adaln(s.name)]
        for spec in lead + rest:
            if (len(spec.shape) <= 1) != small:
                continue
            key = bucket_key(spec.name, len(spec.shape))
            if key != prev:
                total = (total + _BUCKET_ALIGN - 1) // _BUCKET_ALIGN * _BUCKET_ALIGN
                prev = key
            offs[spec.name] = total
            total += ((int(np.prod(spec.shape)) + _ALIGN - 1) // _ALIGN) * _ALIGN
    total = (total + _BUCKET_ALIGN - 1) // _BUCKET_ALIGN * _BUCKET_ALIGN
    return offs, total


def bucket_ranges(table, offs, total):
    """[(key, lo, hi)] in flat order: maximal runs of one bucket key; the ranges tile [0, total) ("rest" appears several times:
    front end, mixer maps; "small" is the tail)."""
    items = sorted(((offs[s.name], bucket_key(s.name, len(s.shape))) for s in table if not s.buffer))
    out = []
    for o, key in items:
        if not out or out[-1][0] != key:
            if out:
                out[-1][2] = o
            out.append([key, o, total])
    return [tuple(r) for r in out]


class _Node(nn.Module):
    """Structural container: exists only so that state_dict() keys mirror the reference module tree."""
    pass


def _ctor_draw(spec: ParamSpec, t: torch.Tensor) -> None:
    """Replays the random draw torch makes when the reference constructs the layer (nn.Linear / nn.Conv2d
    reset_parameters): values are overwritten later, but the RNG stream must advance identically."""
    if spec.ctor in ("linear_w", "conv_w"):
        nn.init.kaiming_uniform_(t.view(t.shape[0], -1) if t.dim() > 2 else t, a=math.sqrt(5))
    elif spec.ctor in ("linear_b", "conv_b"):
        bound = 1.0 / math.sqrt(spec.fan_in) if spec.fan_in > 0 else 0.0
        nn.init.uniform_(t, -bound, bound)
    elif spec.ctor == "ones":
        t.fill_(1.0)
    else:
        t.zero_()


class DiT(nn.Module):
    def __init__(self, input_size: int = 32, patch_size: int = 2, in_channels: int = 4, dim: int = 1152, depth: int = 28,
                 head_dim: int = 64, multiple_of: int = 256, caption_channels: int = 1024, pos_interp_scale: float = 1.0,
                 norm_eps: float = 1e-6, depth_init: bool = True, qkv_multipliers=(1.0,), ffn_multipliers=(4.0,),
                 use_patch_mixer: bool = True, patch_mixer_depth: int = 4, patch_mixer_dim: int = 512,
                 patch_mixer_qkv_ratio: float = 1.0, patch_mixer_mlp_ratio: float = 1.0, use_bias: bool = True,
                 num_experts: int = 8, expert_capacity: float = 1, experts_every_n: int = 2):
        super().__init__()
        if not use_patch_mixer:
            raise NotImplementedError("use_patch_mixer=False is not used by any reference configuration")
        self.config = DiTConfig(input_size, patch_size, in_channels, dim, depth, head_dim, multiple_of, caption_channels,
                                pos_interp_scale, norm_eps, depth_init, tuple(float(v) for v in qkv_multipliers),
                                tuple(float(v) for v in ffn_multipliers), use_patch_mixer, patch_mixer_depth,
                                patch_mixer_dim, patch_mixer_qkv_ratio, patch_mixer_mlp_ratio, use_bias, num_experts,
                                expert_capacity, experts_every_n)
        self.input_size, self.in_channels, self.out_channels = input_size, in_channels, in_channels
        self.patch_size, self.head_dim, self.pos_interp_scale = patch_size, head_dim, pos_interp_scale
        self.use_patch_mixer = use_patch_mixer
        self.base_size = input_size // patch_size
        self._table: List[ParamSpec] = param_table(self.config)
        self._engine: Optional[DiTEngine] = None
        self._flat = None
        self._shadow_version = -1
        self._grad_anchor = None
        self._build_and_init()

    # ------------------------------------------------------------------------------------------ construction
    def _owner(self, dotted: str):
        """Create (or fetch) the container chain for 'a.b.c.weight' and return (module c, 'weight')."""
        parts = dotted.split(".")
        mod = self
        for name in parts[:-1]:
            nxt = mod._modules.get(name)
            if nxt is None:
                nxt = _Node()
                mod.add_module(name, nxt)
            mod = nxt
        return mod, parts[-1]

    def _build_and_init(self) -> None:
        cfg = self.config
        tensors: Dict[str, torch.Tensor] = {}
        for spec in self._table:                      # construction order == table order for every drawing layer
            t = torch.empty(spec.shape, dtype=torch.float32)
            _ctor_draw(spec, t)
            tensors[spec.name] = t
        by_name = {s.name: s for s in self._table}
        # ---- reference initialize_weights (dit.py:577-627), in its exact draw order
        for spec in self._table:                      # (a) xavier on every nn.Linear weight, zero biases
            if spec.ctor == "linear_w":
                nn.init.xavier_uniform_(tensors[spec.name])
            elif spec.ctor == "linear_b":
                tensors[spec.name].zero_()
        grid = cfg.input_size // cfg.patch_size
        tensors["pos_embed"].copy_(torch.from_numpy(sincos_table(cfg.dim, grid, cfg.pos_interp_scale)).float().unsqueeze(0))
        w = tensors["x_embedder.proj.weight"]
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))                                   # (b)
        for nm in ("t_embedder.mlp.0", "t_embedder.mlp.2", "pooled_y_emb_process.fc1", "pooled_y_emb_process.fc2",
                   "y_embedder.y_proj.fc1", "y_embedder.y_proj.fc2"):                   # (c)
            nn.init.normal_(tensors[nm + ".weight"], std=0.02)
        mixer, backbone = plan_blocks(cfg)

        def trunc(name):
            kind = by_name[name].init
            std = kind[1] if kind[0] == "trunc" else 0.02
            nn.init.trunc_normal_(tensors[name], mean=0.0, std=std)

        def block_init(bp):
            q = bp.name
            for nm in (".attn.qkv.weight", ".attn.proj.weight", ".cross_attn.q_linear.weight",
                       ".cross_attn.kv_linear.weight", ".cross_attn.proj.weight"):
                trunc(q + nm)
            if bp.moe:
                for nm in (".mlp.gate.weight", ".mlp.w1", ".mlp.w2"):
                    trunc(q + nm)
            else:
                for nm in (".mlp.w1.weight", ".mlp.w2.weight", ".mlp.w3.weight"):
                    trunc(q + nm)

        for bp in backbone:                                                              # (d) backbone first ...
            block_init(bp)
        for bp in mixer:                                                                 # (e) ... then the mixer
            block_init(bp)
        for nm in ("attn.qkv", "attn.proj", "mlp.w1", "mlp.w2", "mlp.w3"):               # (g) caption block, std 0.02
            nn.init.trunc_normal_(tensors["y_emb_preprocess." + nm + ".weight"], mean=0.0, std=0.02)
        for spec in self._table:                                                         # (f, h) zero-initialised outputs
            if spec.init == ("zeros",) and spec.ctor == "linear_w":
                tensors[spec.name].zero_()
        # ---- register as parameters / buffers (state_dict layout of the reference)
        for spec in self._table:
            owner, leaf = self._owner(spec.name)
            if spec.buffer:
                owner.register_buffer(leaf, tensors[spec.name])
            else:
                owner.register_parameter(leaf, nn.Parameter(tensors[spec.name]))

    # ------------------------------------------------------------------------------------------ flat storage
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flat = None            # storage moved: re-flatten (eagerly on a GPU so optimisers see the final tensors)
        self._engine = None
        first = next(self.parameters())
        if first.device.type == "cuda":
            self._ensure_flat()
        return out

    def _ensure_flat(self) -> None:
        """Move every parameter into one flat fp32 device buffer (keeping values and Parameter identity), create the
        bf16 shadow and the fp32 gradient buffer with the same layout, and bind the engine."""
        params = dict(self.named_parameters())
        first = next(iter(params.values()))
        if self._flat is not None and self._flat["p"].device == first.device and self._engine is not None:
            return
        dev = first.device
        if dev.type != "cuda":
            raise RuntimeError("micro_diffusion_amd.DiT runs only on an AMD GPU (HIP kernels); call .to('cuda') first. "
                               "There is no CPU fallback.")
        offs, total = flat_layout(self._table)
        flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        flat_s = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        P, G, S = {}, {}, {}
        for spec in self._table:
            if spec.buffer:
                continue
            n, o = int(np.prod(spec.shape)), offs[spec.name]
            view = flat_p[o:o + n].view(spec.shape)
            par = params[spec.name]
            view.copy_(par.detach().to(device=dev, dtype=torch.float32))
            par.data = view                     # same Parameter object, storage now inside the flat buffer
            par.grad = None
            P[spec.name] = view
            G[spec.name] = flat_g[o:o + n].view(spec.shape)
            S[spec.name] = flat_s[o:o + n].view(spec.shape)
        buf = {k: v for k, v in self.named_buffers()}
        self._plist = [params[s.name] for s in self._table if not s.buffer]
        self._flat = {"p": flat_p, "g": flat_g, "s": flat_s, "offs": offs, "total": total, "P": P, "G": G, "S": S,
                      "buckets": bucket_ranges(self._table, offs, total)}
        self._engine = DiTEngine(self.config, P, S, G, buf)
        self._shadow_version = -1
        self._grad_anchor = torch.zeros(1, device=dev, requires_grad=True)

    def _param_version(self) -> int:
        return sum(p._version for p in self._plist)

    def refresh_shadow(self, force: bool = False) -> None:
        """bf16 shadow := fp32 masters whenever a master changed (optimizer step, load_state_dict, manual edit):
        detected through the parameters' autograd version counters, which every in-place update bumps."""
        f = self._flat
        ver = self._param_version()
        if (force or ver != self._shadow_version) and getattr(self, "shadow_is_authoritative", False):
            # (`force` too: callers that write the flat master buffer directly -- EMA swap, sync_replicas -- do not bump the
            # Parameters' version counters, and a forced cast from stale masters is the same corruption: ADVICE r4)
            # sharded optimiser: this rank's fp32 masters of the OTHER ranks' chunks are stale and the bf16 shadow (all-gathered)
            # is the only complete copy of the weights -- re-deriving it from the masters would silently corrupt them
            raise RuntimeError("the bf16 shadow would be re-derived from the fp32 masters while the sharded optimiser holds stale "
                               "masters of foreign chunks (a parameter was modified in place, or swap_ema / sync_replicas ran): call "
                               "Trainer.consolidate() first (it all-gathers masters and moments)")
        if force or ver != self._shadow_version:
            hip.check(hip.lib().md_cast_f32_bf16(f["p"].data_ptr(), f["s"].data_ptr(), f["total"], None,
                                                 torch.cuda.current_stream().cuda_stream), "md_cast_f32_bf16")
            self._shadow_version = ver

    def mark_shadow_fresh(self) -> None:
        """Called by the fused optimiser, which writes the masters AND the shadow in one pass."""
        self._shadow_version = self._param_version()

    @property
    def engine(self) -> DiTEngine:
        self._ensure_flat()
        return self._engine

    def flat_buffers(self):
        self._ensure_flat()
        return self._flat

    def attach_grads(self) -> None:
        """Make every parameter's .grad a view of the flat fp32 gradient buffer (zeroing it if grads were None)."""
        f = self._flat
        if self._plist[0].grad is None or self._plist[-1].grad is None:
            f["g"].zero_()
            for spec, p in zip((s for s in self._table if not s.buffer), self._plist):
                p.grad = f["G"][spec.name]

    # ------------------------------------------------------------------------------------------ forward
    def forward_without_cfg(self, x, t, y, mask_ratio: float = 0, mask_noise: Optional[torch.Tensor] = None, **kwargs):
        self._ensure_flat()
        self.refresh_shadow()
        self.h = x.shape[-2] // self.patch_size
        self.w = x.shape[-1] // self.patch_size
        B = x.shape[0]
        if mask_ratio > 0 and mask_noise is None:
            mask_noise = torch.rand(B, self.h * self.w, device=x.device)     # same draw as utils.py:390
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        xin = x.detach().to(torch.float32).contiguous()
        yin = y.detach()
        if yin.dtype not in (torch.float16, torch.float32):
            yin = yin.float()
        yin = yin.contiguous()
        if need_grad:
            sample, mask = _DiTFunction.apply(self, self._grad_anchor, xin, t.detach(), yin, float(mask_ratio), mask_noise)
        else:
            tape = self._engine.forward(xin, t.detach(), yin, mask_ratio=float(mask_ratio), mask_noise=mask_noise)
            sample, mask = self._engine.sample_image(tape), tape.mask
        return {"sample": sample, "mask": mask}

    def forward_with_cfg(self, x, t, y, cfg: float = 1.0, mask_ratio: float = 0, **kwargs):
        """Classifier-free guidance by batch doubling with zeroed captions (reference dit.py:542-550)."""
        x2 = torch.cat([x, x], 0)
        y2 = torch.cat([y, torch.zeros_like(y)], 0)
        t2 = torch.cat([t, t], 0) if len(t) != 1 else t
        eps = self.forward_without_cfg(x2, t2, y2, mask_ratio, **kwargs)["sample"]
        cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
        return {"sample": uncond + cfg * (cond - uncond)}

    def forward(self, x, t, y, cfg: float = 1.0, **kwargs):
        if cfg != 1.0:
            return self.forward_with_cfg(x, t, y, cfg, **kwargs)
        return self.forward_without_cfg(x, t, y, **kwargs)

    def unpatchify(self, x: torch.Tensor) -> torch.Tensor:
        """[B, T, p*p*C] with (ph, pw, c) channel order -> [B, C, H, W] (API parity with dit.py:566-575; glue)."""
        c, p = self.out_channels, self.patch_size
        g = int(x.shape[1] ** 0.5)
        assert g * g == x.shape[1]
        return x.reshape(x.shape[0], g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], c, g * p, g * p)


class _DiTFunction(torch.autograd.Function):
    """The whole network as one autograd node.  Parameter gradients are accumulated by the engine directly into
    the flat fp32 gradient buffer (every parameter's .grad is a view of it), not returned through autograd."""

    @staticmethod
    def forward(ctx, module: DiT, anchor, x, t, y, mask_ratio, mask_noise):
        eng = module._engine
        tape = eng.forward(x, t, y, mask_ratio=mask_ratio, mask_noise=mask_noise, record_tape=True)
        ctx.module, ctx.tape = module, tape
        img = eng.sample_image(tape)
        mask = tape.mask
        if mask is not None:
            ctx.mark_non_differentiable(mask)
        return img, mask

    @staticmethod
    def backward(ctx, dimg, dmask):
        module, tape = ctx.module, ctx.tape
        eng = module._engine
        cfg = module.config
        module.attach_grads()
        # d(sample)/d(token output): gather the image grad into kept-token rows, (ph, pw, c) order
        B, C, H, W, p = tape.B, cfg.in_channels, tape.H, tape.W, cfg.patch_size
        g = H // p
        dtok_all = dimg.contiguous().view(B, C, g, p, g, p).permute(0, 2, 4, 3, 5, 1).reshape(B * g * g, p * p * C)
        if tape.keep_rows is not None:
            dtok_all = dtok_all[tape.keep_rows.long()]
        eng.backward(tape, dtok_all.to(torch.bfloat16).contiguous())
        ctx.tape = None
        return None, torch.zeros_like(module._grad_anchor), None, None, None, None, None


# ---------------------------------------------------------------------------------------------- model zoo
def MicroDiT_Tiny_2(caption_channels: int = 1024, qkv_ratio=(0.5, 1.0), mlp_ratio=(0.5, 4.0), pos_interp_scale: float = 1.0,
                    input_size: int = 32, num_experts: int = 8, expert_capacity: float = 2.0, experts_every_n: int = 2,
                    in_channels: int = 4, **kwargs) -> DiT:
    """Reference dit.py:630-668 (16 layers, d=512, head_dim 32, mixer 4x512: Identity mixer maps)."""
    depth = 16
    return DiT(input_size=input_size, patch_size=2, in_channels=in_channels, dim=512, depth=depth, head_dim=32,
               multiple_of=256, caption_channels=caption_channels, pos_interp_scale=pos_interp_scale, norm_eps=1e-6,
               depth_init=True, qkv_multipliers=np.linspace(qkv_ratio[0], qkv_ratio[1], num=depth, dtype=float),
               ffn_multipliers=np.linspace(mlp_ratio[0], mlp_ratio[1], num=depth, dtype=float), use_patch_mixer=True,
               patch_mixer_depth=4, patch_mixer_dim=512, patch_mixer_qkv_ratio=1.0, patch_mixer_mlp_ratio=4.0,
               use_bias=False, num_experts=num_experts, expert_capacity=expert_capacity, experts_every_n=experts_every_n,
               **kwargs)


def MicroDiT_XL_2(caption_channels: int = 1024, qkv_ratio=(0.5, 1.0), mlp_ratio=(0.5, 4.0), pos_interp_scale: float = 1.0,
                  input_size: int = 32, num_experts: int = 8, expert_capacity: float = 2.0, experts_every_n: int = 2,
                  in_channels: int = 4, **kwargs) -> DiT:
    """Reference dit.py:671-709 (28 layers, d=1024, head_dim 64, mixer 6x768; 1,165,442,320 parameters)."""
    depth = 28
    return DiT(input_size=input_size, patch_size=2, in_channels=in_channels, dim=1024, depth=depth, head_dim=64,
               multiple_of=256, caption_channels=caption_channels, pos_interp_scale=pos_interp_scale, norm_eps=1e-6,
               depth_init=True, qkv_multipliers=np.linspace(qkv_ratio[0], qkv_ratio[1], num=depth, dtype=float),
               ffn_multipliers=np.linspace(mlp_ratio[0], mlp_ratio[1], num=depth, dtype=float), use_patch_mixer=True,
               patch_mixer_depth=6, patch_mixer_dim=768, patch_mixer_qkv_ratio=1.0, patch_mixer_mlp_ratio=4.0,
               use_bias=False, num_experts=num_experts, expert_capacity=expert_capacity, experts_every_n=experts_every_n,
               **kwargs)


def MicroDiT_Tiny(caption_channels: int = 1024, pos_interp_scale: float = 1.0, input_size: int = 32, in_channels: int = 4,
                  **kwargs) -> DiT:
    """BASELINE.json configs[0] 'MicroDiT-Tiny (2 layers, d=256)' (SURVEY.md §8d); not a reference zoo entry."""
    return DiT(input_size=input_size, patch_size=2, in_channels=in_channels, dim=256, depth=2, head_dim=32, multiple_of=256,
               caption_channels=caption_channels, pos_interp_scale=pos_interp_scale, qkv_multipliers=[1.0],
               ffn_multipliers=[4.0], patch_mixer_depth=2, patch_mixer_dim=128, patch_mixer_mlp_ratio=4.0, use_bias=False,
               num_experts=8, expert_capacity=2.0, experts_every_n=2, **kwargs)
