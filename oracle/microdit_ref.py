"""ORACLE — test infrastructure only.  CPU fp32 restatement of the reference MicroDiT training hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; the product
package (`micro_diffusion_amd`) never does.

What it restates (reference = SonyResearch/micro_diffusion @ 2025-02-16, paths relative to /root/reference):
  * DiT.forward_without_cfg                          micro_diffusion/models/dit.py:455-519
  * DiTBlock / FeedForward / FeedForwardECMoe        dit.py:63-148, 232-239
  * AttentionBlockPromptEmbedding                    dit.py:12-60
  * SelfAttention / CrossAttention / T2IFinalLayer   micro_diffusion/models/utils.py:81-240
  * TimestepEmbedder / CaptionProjection / Mlp       utils.py:34-68, 243-318
  * sin-cos position table                           utils.py:330-379
  * get_mask / mask_out_token / unmask_tokens        utils.py:382-426
  * EDM preconditioning + masked loss                micro_diffusion/models/model.py:144-210
  * clip_grad_norm_ / AdamW / LR schedules           train.py:39-43,85-86; configs/*.yaml (torch / Composer
                                                     semantics, SURVEY.md Appendix C items 1,3,5)

The model is a pure function of a `state_dict` (the reference's 478-key layout) so the same
weights can drive the reference (oracle/gen_golden.py), this restatement, and the HIP engine.

Parity pinning: the reference ships no tests or golden vectors.  This restatement is pinned by
`tests/golden/*.npz`, produced by running the *unmodified reference code* in the build container
(oracle/gen_golden.py) and checked by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ------------------------------------------------------------------------------------------- architecture
@dataclass
class RefConfig:
    """Constructor arguments of reference `DiT` (dit.py:277-301)."""
    input_size: int = 32
    patch_size: int = 2
    in_channels: int = 4
    dim: int = 1024
    depth: int = 28
    head_dim: int = 64
    multiple_of: int = 256
    caption_channels: int = 1024
    pos_interp_scale: float = 1.0
    norm_eps: float = 1e-6
    depth_init: bool = True
    qkv_multipliers: Sequence[float] = (1.0,)
    ffn_multipliers: Sequence[float] = (4.0,)
    patch_mixer_depth: int = 6
    patch_mixer_dim: int = 768
    patch_mixer_qkv_ratio: float = 1.0
    patch_mixer_mlp_ratio: float = 4.0
    use_bias: bool = False
    num_experts: int = 8
    expert_capacity: float = 2.0
    experts_every_n: int = 2


def xl2_config(input_size=32, pos_interp_scale=1.0, caption_channels=1024) -> RefConfig:
    """MicroDiT_XL_2 (dit.py:671-709)."""
    depth = 28
    return RefConfig(input_size=input_size, dim=1024, depth=depth, head_dim=64, caption_channels=caption_channels,
                     pos_interp_scale=pos_interp_scale,
                     qkv_multipliers=tuple(np.linspace(0.5, 1.0, num=depth, dtype=float)),
                     ffn_multipliers=tuple(np.linspace(0.5, 4.0, num=depth, dtype=float)),
                     patch_mixer_depth=6, patch_mixer_dim=768, patch_mixer_mlp_ratio=4.0)


def tiny_config(input_size=32) -> RefConfig:
    """BASELINE.json configs[0] 'MicroDiT-Tiny (2 layers, d=256)' (SURVEY.md §8d)."""
    return RefConfig(input_size=input_size, dim=256, depth=2, head_dim=32, qkv_multipliers=(1.0,),
                     ffn_multipliers=(4.0,), patch_mixer_depth=2, patch_mixer_dim=128, patch_mixer_mlp_ratio=4.0)


def tiny512_config() -> RefConfig:
    """configs/res_512_*.yaml geometry on the Tiny widths: 64 x 64 latents (T = 1024 tokens), pos_interp_scale = 2
    (BASELINE.json configs[3..4] shapes)."""
    c = tiny_config(input_size=64)
    c.pos_interp_scale = 2.0
    return c


def micro_config() -> RefConfig:
    """Small config exercising branches XL/2 and Tiny do not: patch_mixer_dim == dim (Identity maps,
    dit.py:389-392), use_bias=True, split multiplier lists, 4 experts, narrow captions."""
    return RefConfig(input_size=16, dim=128, depth=4, head_dim=32, caption_channels=64, multiple_of=64,
                     qkv_multipliers=(0.5, 1.0), ffn_multipliers=(1.0, 4.0), patch_mixer_depth=2,
                     patch_mixer_dim=128, patch_mixer_mlp_ratio=2.0, use_bias=True, num_experts=4,
                     expert_capacity=2.0)


def _round_up(v: int, m: int) -> int:
    return m * ((v + m - 1) // m)


@dataclass
class BlockSpec:
    prefix: str
    dim: int
    attn_hidden: int
    xattn_hidden: int
    moe: bool
    ffn_hidden: int
    init_std: float
    heads: int = field(init=False)
    xheads: int = field(init=False)


def block_specs(cfg: RefConfig):
    """Widths of every DiTBlock (dit.py:192-196, 81-82, 119, 346-353, 394-418)."""
    hd = cfg.head_dim

    def one(prefix, dim, qkv_ratio, mlp_ratio, moe, std):
        h = dim if qkv_ratio == 1 else (2 * hd) * ((int(dim * qkv_ratio) + 2 * hd - 1) // (2 * hd))
        hid = int(dim * mlp_ratio)
        f = _round_up(hid, cfg.multiple_of) if moe else _round_up(int(2 * hid / 3), cfg.multiple_of)
        s = BlockSpec(prefix, dim, h, dim, moe, f, std)
        s.heads, s.xheads = h // hd, dim // hd
        return s

    mixer = []
    for i in range(cfg.patch_mixer_depth):
        moe = i >= 1 and (i + 1) % cfg.experts_every_n == 0
        mixer.append(one(f"patch_mixer.{i}", cfg.patch_mixer_dim, cfg.patch_mixer_qkv_ratio,
                         cfg.patch_mixer_mlp_ratio, moe, 0.02 / (2 * cfg.depth) ** 0.5))
    if len(cfg.ffn_multipliers) == cfg.depth:
        qr, mr = list(cfg.qkv_multipliers), list(cfg.ffn_multipliers)
    else:
        per = cfg.depth // len(cfg.ffn_multipliers)
        qr = [m for m in cfg.qkv_multipliers for _ in range(per)]
        mr = [m for m in cfg.ffn_multipliers for _ in range(per)]
    blocks = []
    for i in range(cfg.depth):
        moe = i < cfg.depth - 1 and (i + 1) % cfg.experts_every_n == 0
        std = 0.02 / (2 * (i + 1)) ** 0.5 if cfg.depth_init else 0.02 / (2 * cfg.depth) ** 0.5
        blocks.append(one(f"blocks.{i}", cfg.dim, qr[i], mr[i], moe, std))
    return mixer, blocks


def sincos_pos_embed(dim: int, grid: int, pos_interp_scale: float, base_size: int) -> np.ndarray:
    """utils.py:330-379.  Token (row i, col j): first dim/2 channels encode the column coordinate, last dim/2
    the row coordinate; each half = [sin(p*w), cos(p*w)], w_k = 10000^(-k/(dim/4))."""
    axis = np.arange(grid, dtype=np.float32) / (grid / base_size) / pos_interp_scale
    col = np.tile(axis[None, :], (grid, 1)).reshape(-1)   # varies fastest
    row = np.tile(axis[:, None], (1, grid)).reshape(-1)
    quarter = dim // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)

    def enc(p):
        a = p[:, None] * omega[None, :]
        return np.concatenate([np.sin(a), np.cos(a)], axis=1)

    return np.concatenate([enc(col), enc(row)], axis=1)


# ------------------------------------------------------------------------------------------- building blocks
def _lin(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln_w(sd: SD, name: str, x: Tensor, eps: float) -> Tensor:
    """create_norm('layernorm'): weight only, no bias (utils.py:71-74)."""
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], None, eps)


def _ln_np(x: Tensor, eps: float) -> Tensor:
    """create_norm('np_layernorm') (utils.py:75-76)."""
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def _gelu_tanh(x):
    return F.gelu(x, approximate="tanh")


def _attention(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """q [B,Sq,H*hd], k/v [B,Sk,H*hd] -> [B,Sq,H*hd]; softmax(q k^T / sqrt(hd)) v, no mask (utils.py:127-132)."""
    B, Sq, hid = q.shape
    hd = hid // heads
    qh = q.view(B, Sq, heads, hd).transpose(1, 2)
    kh = k.view(B, -1, heads, hd).transpose(1, 2)
    vh = v.view(B, -1, heads, hd).transpose(1, 2)
    att = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, Sq, hid)


def self_attention(sd: SD, pre: str, x: Tensor, heads: int, eps: float) -> Tensor:
    """utils.py:178-197: qkv split as (3, heads, hd); LN over the whole hidden width on q and k."""
    B, N, _ = x.shape
    qkv = _lin(sd, pre + ".qkv", x)
    hid = qkv.shape[-1] // 3
    q, k, v = qkv.view(B, N, 3, hid).unbind(2)
    o = _attention(_ln_np(q, eps), _ln_np(k, eps), v, heads)
    return _lin(sd, pre + ".proj", o)


def cross_attention(sd: SD, pre: str, x: Tensor, cond: Tensor, heads: int, eps: float) -> Tensor:
    """utils.py:116-136: kv split as (2, heads, hd)."""
    B = x.shape[0]
    q = _lin(sd, pre + ".q_linear", x)
    kv = _lin(sd, pre + ".kv_linear", cond.reshape(B, -1, cond.shape[-1]))
    hid = q.shape[-1]
    k, v = kv.view(B, -1, 2, hid).unbind(2)
    o = _attention(_ln_np(q, eps), _ln_np(k, eps), v, heads)
    return _lin(sd, pre + ".proj", o)


def swiglu(sd: SD, pre: str, x: Tensor) -> Tensor:
    """dit.py:88-89."""
    return _lin(sd, pre + ".w3", F.silu(_lin(sd, pre + ".w1", x)) * _lin(sd, pre + ".w2", x))


def ec_moe(sd: SD, pre: str, x: Tensor, num_experts: int, capacity: float, return_routing=False):
    """Expert-choice MoE (dit.py:126-143), restated with gather / index_add instead of one-hot matmuls."""
    n, t, d = x.shape
    k = int(capacity * t / num_experts)
    probs = torch.softmax(F.linear(x, sd[pre + ".gate.weight"]), dim=-1)         # [n,t,e]
    g, m = torch.topk(probs.permute(0, 2, 1), k, dim=-1)                           # [n,e,k]
    xin = torch.gather(x.unsqueeze(1).expand(n, num_experts, t, d), 2, m.unsqueeze(-1).expand(-1, -1, -1, d))
    h = F.gelu(torch.einsum("nekd,edf->nekf", xin, sd[pre + ".w1"]))              # exact erf GELU (dit.py:124)
    h = torch.einsum("nekf,efd->nekd", h, sd[pre + ".w2"]) * g.unsqueeze(-1)
    out = torch.zeros_like(x)
    out.scatter_add_(1, m.reshape(n, num_experts * k, 1).expand(-1, -1, d), h.reshape(n, num_experts * k, d))
    if return_routing:
        return out, m, g
    return out


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def dit_block(sd: SD, spec: BlockSpec, cfg: RefConfig, x: Tensor, y: Tensor, c: Tensor, taps: Optional[dict] = None) -> Tensor:
    """dit.py:232-239.  `taps` (tests): receives the block output and, for expert-choice blocks, the top-k token indices
    [n, e, k] (so a product run can be fed the oracle's routing and routing flips separated from arithmetic error)."""
    p, eps = spec.prefix, cfg.norm_eps
    mod = _lin(sd, p + ".adaLN_modulation.1", _gelu_tanh(c))
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    x = x + g_a.unsqueeze(1) * self_attention(sd, p + ".attn", modulate(_ln_w(sd, p + ".norm1", x, eps), sh_a, sc_a),
                                              spec.heads, eps)
    x = x + cross_attention(sd, p + ".cross_attn", _ln_w(sd, p + ".norm2", x, eps), y, spec.xheads, eps)
    hin = modulate(_ln_w(sd, p + ".norm3", x, eps), sh_m, sc_m)
    if spec.moe and taps is not None:
        ff, m, _ = ec_moe(sd, p + ".mlp", hin, cfg.num_experts, cfg.expert_capacity, return_routing=True)
        taps["route::" + p] = m
    else:
        ff = ec_moe(sd, p + ".mlp", hin, cfg.num_experts, cfg.expert_capacity) if spec.moe else swiglu(sd, p + ".mlp", hin)
    out = x + g_m.unsqueeze(1) * ff
    if taps is not None:
        taps["out::" + p] = out.detach()
    return out


def get_mask(noise: Tensor, mask_ratio: float):
    """utils.py:382-403 given the uniform `noise` [B, L]; ties resolve to the lowest index first (stable)."""
    B, L = noise.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :len_keep]
    mask = (ids_restore >= len_keep).to(torch.float32)
    return {"mask": mask, "ids_keep": ids_keep, "ids_restore": ids_restore}


def timestep_embedding(t: Tensor, dim: int = 512) -> Tensor:
    """utils.py:266-281: [cos(t f_k), sin(t f_k)], f_k = exp(-ln(1e4) k / (dim/2))."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _mlp_norm(sd: SD, pre: str, x: Tensor, eps: float) -> Tensor:
    """Mlp with act before norm (utils.py:63-68)."""
    return _lin(sd, pre + ".fc2", _ln_w(sd, pre + ".norm", _gelu_tanh(_lin(sd, pre + ".fc1", x)), eps))


def unpatchify(x: Tensor, c: int, p: int) -> Tensor:
    """dit.py:566-575: token vector is ordered (ph, pw, c)."""
    B, T, _ = x.shape
    g = int(round(T ** 0.5))
    return x.view(B, g, g, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(B, c, g * p, g * p)


def dit_forward(sd: SD, cfg: RefConfig, x: Tensor, t: Tensor, y: Tensor, mask_ratio: float = 0.0,
                mask_noise: Optional[Tensor] = None, taps: Optional[dict] = None):
    """DiT.forward_without_cfg (dit.py:455-519).  x [B,C,H,W], t [B], y [B,1,L,Dc] -> sample [B,C,H,W], mask."""
    eps, p = cfg.norm_eps, cfg.patch_size
    mixer, blocks = block_specs(cfg)
    B = x.shape[0]
    tok = F.conv2d(x, sd["x_embedder.proj.weight"], sd.get("x_embedder.proj.bias"), stride=p)
    tok = tok.flatten(2).transpose(1, 2) + sd["pos_embed"]
    temb = _lin(sd, "t_embedder.mlp.2", _gelu_tanh(_lin(sd, "t_embedder.mlp.0", timestep_embedding(t.expand(B)))))
    yy = _mlp_norm(sd, "y_embedder.y_proj", y, eps).squeeze(1)                      # [B,L,D]
    yy = yy + self_attention(sd, "y_emb_preprocess.attn", _ln_w(sd, "y_emb_preprocess.norm1", yy, eps),
                             cfg.dim // cfg.head_dim, eps)
    yy = yy + swiglu(sd, "y_emb_preprocess.mlp", _ln_w(sd, "y_emb_preprocess.norm2", yy, eps))
    c = temb + _mlp_norm(sd, "pooled_y_emb_process", yy.mean(dim=1), eps)
    if taps is not None:
        taps["c"], taps["y"] = c, yy
    h = tok
    if "patch_mixer_map_xin.1.weight" in sd:
        h = _lin(sd, "patch_mixer_map_xin.1", _ln_w(sd, "patch_mixer_map_xin.0", h, eps))
        y_mixer = _lin(sd, "patch_mixer_map_y.1", _ln_w(sd, "patch_mixer_map_y.0", yy, eps))
    else:
        y_mixer = yy
    for spec in mixer:
        h = dit_block(sd, spec, cfg, h, y_mixer, c, taps)
    if taps is not None:
        taps["mixer_out"] = h
    mask = ids_restore = None
    if mask_ratio > 0:
        assert mask_noise is not None, "the oracle takes the mask noise explicitly"
        md = get_mask(mask_noise, mask_ratio)
        mask, ids_restore = md["mask"], md["ids_restore"]
        h = torch.gather(h, 1, md["ids_keep"].unsqueeze(-1).expand(-1, -1, h.shape[-1]))
    if "patch_mixer_map_xout.1.weight" in sd:
        h = _lin(sd, "patch_mixer_map_xout.1", _ln_w(sd, "patch_mixer_map_xout.0", h, eps))
    for spec in blocks:
        h = dit_block(sd, spec, cfg, h, yy, c, taps)
    if taps is not None:
        taps["backbone_out"] = h
    shift, scale = _lin(sd, "final_layer.adaLN_modulation.1", _gelu_tanh(c)).chunk(2, dim=1)
    h = _lin(sd, "final_layer.linear", modulate(_ln_w(sd, "final_layer.norm_final", h, eps), shift, scale))
    if mask_ratio > 0:
        pad = sd["mask_token"].expand(B, ids_restore.shape[1] - h.shape[1], -1)
        h = torch.gather(torch.cat([h, pad], dim=1), 1, ids_restore.unsqueeze(-1).expand(-1, -1, h.shape[-1]))
    return unpatchify(h, cfg.in_channels, p), mask


# ------------------------------------------------------------------------------------------- EDM loss
def edm_loss(sd: SD, cfg: RefConfig, x: Tensor, y: Tensor, rnd_normal: Tensor, eps_noise: Tensor,
             mask_ratio: float, mask_noise: Optional[Tensor], p_mean: float, p_std: float, sigma_data: float = 0.9,
             return_parts: bool = False, taps: Optional[dict] = None):
    """model.py:181-210 with the three random draws passed in (order: randn[B,1,1,1], randn_like(x), rand[B,T])."""
    sigma = (rnd_normal.view(-1, 1, 1, 1) * p_std + p_mean).exp()
    weight = (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2
    xn = x + eps_noise * sigma
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    Fx, mask = dit_forward(sd, cfg, c_in * xn, c_noise.flatten(), y, mask_ratio, mask_noise, taps=taps)
    D = c_skip * xn + c_out * Fx
    if taps is not None:
        taps["F"] = Fx.detach()
    loss = weight * (D - x) ** 2
    if mask_ratio > 0:
        loss = F.avg_pool2d(loss.mean(dim=1), cfg.patch_size).flatten(1)
        keep = 1 - mask
        loss = (loss * keep).sum(dim=1) / keep.sum(dim=1)
    out = loss.mean()
    if return_parts:
        return out, {"F": Fx, "D": D, "mask": mask, "sigma": sigma}
    return out


def latent_diffusion_forward(sd, cfg, batch, rnd_normal, eps_noise, mask_noise, mask_ratio, p_mean, p_std, taps=None):
    """LatentDiffusion.forward for precomputed latents (model.py:104-142): caption drop then fp32 casts."""
    lat = batch["image_latents"].float()
    cond = batch["caption_latents"].float()
    if "drop_caption_mask" in batch:
        cond = cond * batch["drop_caption_mask"].view(-1, 1, 1, 1).float()
    return edm_loss(sd, cfg, lat, cond, rnd_normal, eps_noise, mask_ratio, mask_noise, p_mean, p_std, taps=taps)


# ------------------------------------------------------------------------------------------- optimiser side
def clip_grad_norm(grads: List[Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g||_2 + 1e-6)); in place.  Returns the norm."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
               weight_decay=0.1) -> None:
    """torch.optim.AdamW single-tensor math; `step` counts from 1."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (v.sqrt() / math.sqrt(1 - beta2 ** step)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - beta1 ** step))


def lr_factor(kind: str, step: int, t_warmup: int = 0, t_max: int = 1, alpha_f: float = 0.0, alpha: float = 1.0) -> float:
    """Composer schedulers named in configs/*.yaml (SURVEY.md Appendix C.3 — [memory], sources absent):
    `step` = number of optimiser steps already taken (first batch runs at factor(0))."""
    if kind == "constant":
        return alpha
    if kind == "constant_with_warmup":
        return alpha * min(1.0, step / t_warmup) if t_warmup > 0 else alpha
    if kind == "cosine_with_warmup":
        if step < t_warmup:
            return step / t_warmup
        frac = min(1.0, (step - t_warmup) / max(1, t_max - t_warmup))
        return alpha_f + (1 - alpha_f) * 0.5 * (1 + math.cos(math.pi * frac))
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------- synthetic weights
def state_shapes(cfg: RefConfig, caption_len_unused: int = 77) -> Dict[str, tuple]:
    """Key -> shape of `DiT.state_dict()`, derived from the widths above."""
    D, Dm, p, C = cfg.dim, cfg.patch_mixer_dim, cfg.patch_size, cfg.in_channels
    T = (cfg.input_size // p) ** 2
    s: Dict[str, tuple] = {"pos_embed": (1, T, D), "mask_token": (1, 1, p * p * C)}

    def lin(name, o, i, bias):
        s[name + ".weight"] = (o, i)
        if bias:
            s[name + ".bias"] = (o,)

    s["x_embedder.proj.weight"] = (D, C, p, p)
    s["x_embedder.proj.bias"] = (D,)
    lin("t_embedder.mlp.0", D, 512, True)
    lin("t_embedder.mlp.2", D, D, True)
    for pre, cin in (("y_embedder.y_proj", cfg.caption_channels), ("pooled_y_emb_process", D)):
        lin(pre + ".fc1", D, cin, True)
        s[pre + ".norm.weight"] = (D,)
        lin(pre + ".fc2", D, D, True)
    b = cfg.use_bias
    fcap = _round_up(int(2 * int(D * 4.0) / 3), cfg.multiple_of)
    s["y_emb_preprocess.norm1.weight"] = (D,)
    lin("y_emb_preprocess.attn.qkv", 3 * D, D, b)
    lin("y_emb_preprocess.attn.proj", D, D, b)
    s["y_emb_preprocess.norm2.weight"] = (D,)
    lin("y_emb_preprocess.mlp.w1", fcap, D, b)
    lin("y_emb_preprocess.mlp.w2", fcap, D, b)
    lin("y_emb_preprocess.mlp.w3", D, fcap, b)
    if Dm != D:
        for nm, i, o in (("patch_mixer_map_xin", D, Dm), ("patch_mixer_map_xout", Dm, D), ("patch_mixer_map_y", D, Dm)):
            s[nm + ".0.weight"] = (i,)
            lin(nm + ".1", o, i, b)
    mixer, blocks = block_specs(cfg)
    for sp in mixer + blocks:
        q = sp.prefix
        for n in ("norm1", "norm2", "norm3"):
            s[f"{q}.{n}.weight"] = (sp.dim,)
        lin(q + ".attn.qkv", 3 * sp.attn_hidden, sp.dim, b)
        lin(q + ".attn.proj", sp.dim, sp.attn_hidden, b)
        lin(q + ".cross_attn.q_linear", sp.xattn_hidden, sp.dim, b)
        lin(q + ".cross_attn.kv_linear", 2 * sp.xattn_hidden, sp.dim, b)
        lin(q + ".cross_attn.proj", sp.dim, sp.xattn_hidden, b)
        if sp.moe:
            s[q + ".mlp.w1"] = (cfg.num_experts, sp.dim, sp.ffn_hidden)
            s[q + ".mlp.w2"] = (cfg.num_experts, sp.ffn_hidden, sp.dim)
            s[q + ".mlp.gate.weight"] = (cfg.num_experts, sp.dim)
        else:
            lin(q + ".mlp.w1", sp.ffn_hidden, sp.dim, b)
            lin(q + ".mlp.w2", sp.ffn_hidden, sp.dim, b)
            lin(q + ".mlp.w3", sp.dim, sp.ffn_hidden, b)
        lin(q + ".adaLN_modulation.1", 6 * sp.dim, D, True)
    lin("final_layer.linear", p * p * C, D, True)
    lin("final_layer.adaLN_modulation.1", 2 * D, D, True)
    s["final_layer.norm_final.weight"] = (D,)
    return s


def synth_state_dict(cfg: RefConfig, seed: int) -> SD:
    """Deterministic 'de-zeroed' weights (the reference's zero-init tensors make parity
    tests vacuous).  Every tensor is drawn from its own generator keyed by (seed, index in sorted key order), so
    the same weights can be rebuilt anywhere from the shapes alone."""
    shapes = state_shapes(cfg)
    sd: SD = {}
    for idx, key in enumerate(sorted(shapes)):
        shp = shapes[key]
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if key == "pos_embed":
            grid = cfg.input_size // cfg.patch_size
            sd[key] = torch.from_numpy(sincos_pos_embed(cfg.dim, grid, cfg.pos_interp_scale, grid)).float().unsqueeze(0)
        elif key == "mask_token":
            sd[key] = torch.zeros(shp)
        elif len(shp) == 1 and key.endswith(".weight"):                   # every 1-D weight is a LayerNorm weight
            sd[key] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif key.endswith(".bias"):
            sd[key] = 0.05 * torch.randn(shp, generator=g)
        elif "adaLN_modulation" in key:
            sd[key] = torch.randn(shp, generator=g) * (0.5 / math.sqrt(shp[-1]))
        elif key.endswith("mlp.w1") or key.endswith("mlp.w2"):          # MoE [e, in, out]
            sd[key] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(shp[1]))
        elif key == "x_embedder.proj.weight":
            sd[key] = torch.randn(shp, generator=g) * 0.25
        elif key.endswith("gate.weight"):
            sd[key] = torch.randn(shp, generator=g) * (2.0 / math.sqrt(shp[-1]))
        else:
            sd[key] = torch.randn(shp, generator=g) * (1.0 / math.sqrt(shp[-1]))
    return sd


def synth_batch(cfg: RefConfig, B: int, seed: int, cap_len: int = 77, mask_ratio: float = 0.75):
    """Synthetic latents / captions / noise draws."""
    g = torch.Generator().manual_seed(seed)
    T = (cfg.input_size // cfg.patch_size) ** 2
    lat = (torch.randn(B, cfg.in_channels, cfg.input_size, cfg.input_size, generator=g) * 0.8).half()
    cap = torch.randn(B, 1, cap_len, cfg.caption_channels, generator=g).half()
    drop = (torch.rand(B, generator=g) >= 0.1).float()
    rnd = torch.randn(B, 1, 1, 1, generator=g)
    epsn = torch.randn(B, cfg.in_channels, cfg.input_size, cfg.input_size, generator=g)
    mnoise = torch.rand(B, T, generator=g)
    return {"image_latents": lat, "caption_latents": cap, "drop_caption_mask": drop}, rnd, epsn, mnoise


def dezero_state_dict(sd: SD, seed: int = 4321, std: float = 0.02) -> SD:
    """The reference zero-initialises every adaLN / final-layer / caption-block output weight (dit.py:577-627), which makes
    the network contribute almost nothing for the first thousand steps.  The "hot" loss-curve parity run
    (tests/golden/tiny_curve_hot.npz) starts from the same seed-18 initialisation with every all-zero tensor replaced by
    N(0, std^2) noise drawn from its own seeded CPU generator (keys in sorted order)."""
    out = {}
    for i, k in enumerate(sorted(sd)):
        v = sd[k]
        if v.is_floating_point() and v.numel() > 1 and float(v.abs().max()) == 0.0:
            g = torch.Generator().manual_seed(seed + i)
            out[k] = (torch.randn(v.shape, generator=g) * std).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def curve_inputs(cfg: RefConfig, step: int, batch: int = 16, pool: int = 64, pool_seed: int = 77):
    """Deterministic data + noise of step `step` of the 1k-step loss-curve parity run (tests/golden/tiny_curve_1k.npz):
    a fixed pool of `pool` synthetic samples cycled in order, and per-step noise from its own CPU generator."""
    pb, _, _, _ = synth_batch(cfg, pool, pool_seed)
    idx = (torch.arange(batch) + step * batch) % pool
    b = {k: v[idx].clone() for k, v in pb.items()}
    g = torch.Generator().manual_seed(5000 + step)
    T = (cfg.input_size // cfg.patch_size) ** 2
    rnd = torch.randn(batch, 1, 1, 1, generator=g)
    eps = torch.randn(batch, cfg.in_channels, cfg.input_size, cfg.input_size, generator=g)
    mnoise = torch.rand(batch, T, generator=g)
    return b, rnd, eps, mnoise


def edm_sampler(sd: SD, cfg: RefConfig, latents: Tensor, y: Tensor, steps: int, guidance: float = 1.0, sigma_min=0.002,
                sigma_max=80.0, rho=7.0, sigma_data: float = 0.9) -> Tensor:
    """Deterministic EDM Heun sampler with classifier-free guidance (model.py:231-297, dit.py:521-550; S_churn = 0 so
    no noise is injected): fp64 state, network evaluated in fp32 at mask ratio 0."""
    def denoise(x64, sigma):
        x = x64.to(torch.float32)
        s = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1, 1, 1, 1)
        c_skip = sigma_data ** 2 / (s ** 2 + sigma_data ** 2)
        c_out = s * sigma_data / (s ** 2 + sigma_data ** 2).sqrt()
        c_in = 1 / (sigma_data ** 2 + s ** 2).sqrt()
        t = (s.log() / 4).flatten()
        if guidance > 1.0:
            xin = torch.cat([c_in * x, c_in * x], 0)
            yy = torch.cat([y, torch.zeros_like(y)], 0)
            Fx, _ = dit_forward(sd, cfg, xin, t, yy, 0.0)
            cond, uncond = Fx.chunk(2, 0)
            Fx = uncond + guidance * (cond - uncond)
        else:
            Fx, _ = dit_forward(sd, cfg, c_in * x, t, y, 0.0)
        return (c_skip * x + c_out * Fx).to(torch.float64)

    idx = torch.arange(steps, dtype=torch.float64)
    t_steps = (sigma_max ** (1 / rho) + idx / (steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    t_steps = torch.cat([t_steps, torch.zeros(1, dtype=torch.float64)])
    x_next = latents.to(torch.float64) * t_steps[0]
    with torch.no_grad():
        for i in range(steps):
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            x_hat = x_next
            d_cur = (x_hat - denoise(x_hat, t_cur)) / t_cur
            x_next = x_hat + (t_next - t_cur) * d_cur
            if i < steps - 1:
                d_prime = (x_next - denoise(x_next, t_next)) / t_next
                x_next = x_hat + (t_next - t_cur) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next.to(torch.float32)
