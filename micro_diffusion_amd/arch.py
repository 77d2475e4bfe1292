"""Architecture description of MicroDiT: layer widths, parameter table and the weight initialisation.

Everything here is derived from the constructor arguments the reference's `DiT` takes (reference
micro_diffusion/models/dit.py:277-301); the width rules are the ones of dit.py:81-82,119,192-196,346-353,
394-418.  The parameter table lists every tensor of `DiT.state_dict()` (478 entries for
MicroDiT_XL_2) in the reference's registration order, together with the two random draws the
reference makes for it (torch's default layer init at construction, then `initialize_weights`,
dit.py:577-627) so that `torch.manual_seed(s); DiT(...)` yields bit-identical weights.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class DiTConfig:
    input_size: int = 32
    patch_size: int = 2
    in_channels: int = 4
    dim: int = 1152
    depth: int = 28
    head_dim: int = 64
    multiple_of: int = 256
    caption_channels: int = 1024
    pos_interp_scale: float = 1.0
    norm_eps: float = 1e-6
    depth_init: bool = True
    qkv_multipliers: Sequence[float] = (1.0,)
    ffn_multipliers: Sequence[float] = (4.0,)
    use_patch_mixer: bool = True
    patch_mixer_depth: int = 4
    patch_mixer_dim: int = 512
    patch_mixer_qkv_ratio: float = 1.0
    patch_mixer_mlp_ratio: float = 1.0
    use_bias: bool = True
    num_experts: int = 8
    expert_capacity: float = 1
    experts_every_n: int = 2

    @property
    def tokens(self) -> int:
        return (self.input_size // self.patch_size) ** 2

    @property
    def patch_vec(self) -> int:
        return self.patch_size ** 2 * self.in_channels

    @property
    def has_maps(self) -> bool:
        return self.use_patch_mixer and self.patch_mixer_dim != self.dim


def ceil_to(v: int, m: int) -> int:
    return ((v + m - 1) // m) * m


@dataclass
class BlockPlan:
    name: str           # "patch_mixer.3" / "blocks.17"
    dim: int            # residual width
    attn_hidden: int    # q/k/v width of the self-attention
    xattn_hidden: int   # q/k/v width of the cross-attention (== dim: compress_xattn=False everywhere)
    ffn_hidden: int
    moe: bool
    init_std: float
    cond_dim: int       # width of the caption tokens this block attends to
    heads: int = 0
    xheads: int = 0


def plan_blocks(cfg: DiTConfig) -> Tuple[List[BlockPlan], List[BlockPlan]]:
    hd = cfg.head_dim

    def make(name, dim, qkv_ratio, mlp_ratio, moe, std, cond_dim):
        hidden = dim if qkv_ratio == 1 else ceil_to(int(dim * qkv_ratio), 2 * hd)
        inner = int(dim * mlp_ratio)
        f = ceil_to(inner, cfg.multiple_of) if moe else ceil_to(int(2 * inner / 3), cfg.multiple_of)
        return BlockPlan(name, dim, hidden, dim, f, moe, std, cond_dim, hidden // hd, dim // hd)

    mixer: List[BlockPlan] = []
    if cfg.use_patch_mixer:
        std = 0.02 / math.sqrt(2 * cfg.depth)      # depth_init=False, num_layers=depth (dit.py:364-366)
        for i in range(cfg.patch_mixer_depth):
            moe = i >= 1 and (i + 1) % cfg.experts_every_n == 0
            mixer.append(make(f"patch_mixer.{i}", cfg.patch_mixer_dim, cfg.patch_mixer_qkv_ratio,
                              cfg.patch_mixer_mlp_ratio, moe, std, cfg.patch_mixer_dim))
    nq = len(cfg.qkv_multipliers)
    assert nq == len(cfg.ffn_multipliers)
    if nq == cfg.depth:
        qr, mr = list(cfg.qkv_multipliers), list(cfg.ffn_multipliers)
    else:
        assert cfg.depth % nq == 0, "number of blocks should be divisible by number of splits"
        qr = list(np.repeat(np.asarray(cfg.qkv_multipliers, dtype=float), cfg.depth // nq))
        mr = list(np.repeat(np.asarray(cfg.ffn_multipliers, dtype=float), cfg.depth // nq))
    backbone: List[BlockPlan] = []
    for i in range(cfg.depth):
        moe = i < cfg.depth - 1 and (i + 1) % cfg.experts_every_n == 0
        std = 0.02 / math.sqrt(2 * (i + 1)) if cfg.depth_init else 0.02 / math.sqrt(2 * cfg.depth)
        backbone.append(make(f"blocks.{i}", cfg.dim, qr[i], mr[i], moe, std, cfg.dim))
    return mixer, backbone


def caption_ffn_hidden(cfg: DiTConfig) -> int:
    return ceil_to(int(2 * int(cfg.dim * 4.0) / 3), cfg.multiple_of)


# A parameter (or buffer) of the model.  `ctor` describes the RNG draw torch makes when the layer is built
# ("linear_w"/"linear_b"/"conv_w"/"conv_b" consume random numbers; "ones"/"zeros" do not); `init` is what
# initialize_weights finally leaves in it.
@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    ctor: str
    init: Tuple = ("keep",)
    buffer: bool = False
    fan_in: int = 0


def param_table(cfg: DiTConfig) -> List[ParamSpec]:
    D, Dm, p, C, E = cfg.dim, cfg.patch_mixer_dim, cfg.patch_size, cfg.in_channels, cfg.num_experts
    out: List[ParamSpec] = []

    def linear(name, o, i, bias, init=("xavier",)):
        out.append(ParamSpec(name + ".weight", (o, i), "linear_w", init, fan_in=i))
        if bias:
            out.append(ParamSpec(name + ".bias", (o,), "linear_b", ("zeros",), fan_in=i))

    def norm(name, width):
        out.append(ParamSpec(name + ".weight", (width,), "ones"))

    out.append(ParamSpec("pos_embed", (1, cfg.tokens, D), "zeros", ("pos_embed",), buffer=True))
    out.append(ParamSpec("mask_token", (1, 1, cfg.patch_vec), "zeros", buffer=True))
    out.append(ParamSpec("x_embedder.proj.weight", (D, C, p, p), "conv_w", ("xavier_flat",), fan_in=C * p * p))
    out.append(ParamSpec("x_embedder.proj.bias", (D,), "conv_b", ("keep",), fan_in=C * p * p))
    linear("t_embedder.mlp.0", D, 512, True, ("normal", 0.02))
    linear("t_embedder.mlp.2", D, D, True, ("normal", 0.02))
    # Mlp(fc1, act, norm, fc2): the norm module is built by the caller BEFORE the Mlp (dit.py:321-326) but is
    # registered between fc1 and fc2 (utils.py:58-61); it draws nothing, so only the key order matters.
    linear("y_embedder.y_proj.fc1", D, cfg.caption_channels, True, ("normal", 0.02))
    norm("y_embedder.y_proj.norm", D)
    linear("y_embedder.y_proj.fc2", D, D, True, ("normal", 0.02))
    b = cfg.use_bias
    fc = caption_ffn_hidden(cfg)
    norm("y_emb_preprocess.norm1", D)
    linear("y_emb_preprocess.attn.qkv", 3 * D, D, b, ("trunc", 0.02))
    linear("y_emb_preprocess.attn.proj", D, D, b, ("zeros",))
    norm("y_emb_preprocess.norm2", D)
    linear("y_emb_preprocess.mlp.w1", fc, D, b, ("trunc", 0.02))
    linear("y_emb_preprocess.mlp.w2", fc, D, b, ("trunc", 0.02))
    linear("y_emb_preprocess.mlp.w3", D, fc, b, ("zeros",))
    linear("pooled_y_emb_process.fc1", D, D, True, ("normal", 0.02))
    norm("pooled_y_emb_process.norm", D)
    linear("pooled_y_emb_process.fc2", D, D, True, ("normal", 0.02))

    mixer, backbone = plan_blocks(cfg)

    def block(bp: BlockPlan):
        q = bp.name
        norm(q + ".norm1", bp.dim)
        linear(q + ".attn.qkv", 3 * bp.attn_hidden, bp.dim, b, ("trunc", 0.02))
        linear(q + ".attn.proj", bp.dim, bp.attn_hidden, b, ("trunc", bp.init_std))
        linear(q + ".cross_attn.q_linear", bp.xattn_hidden, bp.dim, b, ("trunc", 0.02))
        linear(q + ".cross_attn.kv_linear", 2 * bp.xattn_hidden, bp.cond_dim if False else bp.dim, b, ("trunc", 0.02))
        linear(q + ".cross_attn.proj", bp.dim, bp.xattn_hidden, b, ("trunc", bp.init_std))
        norm(q + ".norm2", bp.dim)
        norm(q + ".norm3", bp.dim)
        if bp.moe:
            out.append(ParamSpec(q + ".mlp.w1", (E, bp.dim, bp.ffn_hidden), "ones", ("trunc", 0.02)))
            out.append(ParamSpec(q + ".mlp.w2", (E, bp.ffn_hidden, bp.dim), "ones", ("trunc", bp.init_std)))
            linear(q + ".mlp.gate", E, bp.dim, False, ("trunc", 0.02))
        else:
            linear(q + ".mlp.w1", bp.ffn_hidden, bp.dim, b, ("trunc", 0.02))
            linear(q + ".mlp.w2", bp.ffn_hidden, bp.dim, b, ("trunc", bp.init_std))
            linear(q + ".mlp.w3", bp.dim, bp.ffn_hidden, b, ("trunc", bp.init_std))
        linear(q + ".adaLN_modulation.1", 6 * bp.dim, D, True, ("zeros",))

    for bp in mixer:
        block(bp)
    if cfg.has_maps:
        for nm, i, o in (("patch_mixer_map_xin", D, Dm), ("patch_mixer_map_xout", Dm, D), ("patch_mixer_map_y", D, Dm)):
            norm(nm + ".0", i)
            linear(nm + ".1", o, i, b)
    for bp in backbone:
        block(bp)
    linear("final_layer.linear", cfg.patch_vec, D, True, ("zeros",))
    linear("final_layer.adaLN_modulation.1", 2 * D, D, True, ("zeros",))
    norm("final_layer.norm_final", D)
    return out


def sincos_table(dim: int, grid: int, pos_interp_scale: float) -> np.ndarray:
    """Fixed 2-D sin-cos position table [grid*grid, dim] (reference utils.py:330-379 with base_size == grid, as
    DiT always calls it, dit.py:591-596): per token (row i, col j) the first dim/2 channels encode j, the last
    dim/2 encode i, each as [sin(pos * w_k), cos(pos * w_k)], w_k = 10000^(-k / (dim/4))."""
    assert dim % 4 == 0
    coord = np.arange(grid, dtype=np.float32) / np.float32(1.0) / pos_interp_scale   # grid / base_size == 1
    jj, ii = np.meshgrid(coord, coord)                # jj[i, j] = coord[j], ii[i, j] = coord[i]
    freq = 1.0 / 10000 ** (np.arange(dim // 4, dtype=np.float64) / (dim / 4.0))
    halves = []
    for pos in (jj.reshape(-1), ii.reshape(-1)):
        ang = np.outer(pos, freq)
        halves.append(np.concatenate([np.sin(ang), np.cos(ang)], axis=1))
    return np.concatenate(halves, axis=1)


def is_block_adaln(name: str) -> bool:
    """`patch_mixer.<i>.adaLN_modulation.1.{weight,bias}` / `blocks.<i>.adaLN_modulation.1.{weight,bias}` (dit.py:222-225): the
    modulation Linear of every DiT block reads the SAME input, gelu(c), so the flat layout keeps all their weights (and all their
    biases) contiguous, in forward order, and the engine computes the modulation of every block with ONE GEMM per forward."""
    top = name.split(".")
    return (len(top) == 5 and top[0] in ("blocks", "patch_mixer") and top[1].isdigit() and top[2] == "adaLN_modulation" and top[3] == "1"
            and top[4] in ("weight", "bias"))


def adaln_order(name: str):
    """Sort key of the block adaLN tensors inside their region: mixer blocks, then backbone blocks, by index (= forward order)."""
    top = name.split(".")
    return (0 if top[0] == "patch_mixer" else 1, int(top[1]))


def bucket_key(name: str, ndim: int = 2) -> str:
    """Data-parallel bucket a parameter belongs to: one per DiT block ("blocks.17", "patch_mixer.3"), "final_layer", "rest"
    (embedders, caption block, mixer maps) -- the segments whose backward finishes together (engine.backward's on_segment
    hand-off) --, "adaln.m" / "adaln.b" (the modulation weights of all mixer / all backbone blocks, contiguous at the head of the flat
    buffers and needed together by the first GEMM after the condition vector in the forward; "adaln.b" is complete, and exchanged,
    when the backward of the FIRST backbone block is done -- the mixer's backward still covers it --, "adaln.m" with the last
    segment) and "small" for every one-dimensional tensor (biases, LayerNorm weights): those live in one region at the end of
    the flat buffers, are exchanged as one all-reduce and updated by every rank (the engine reads them from the fp32 masters,
    so every replica must hold them exactly; the sharded optimiser step only owns slices of the matrix-shaped buckets)."""
    if ndim <= 1:
        return "small"
    top = name.split(".")
    if is_block_adaln(name):
        return "adaln.m" if top[0] == "patch_mixer" else "adaln.b"
    if top[0] in ("blocks", "patch_mixer") and len(top) > 1 and top[1].isdigit():
        return ".".join(top[:2])
    return "final_layer" if top[0] == "final_layer" else "rest"
