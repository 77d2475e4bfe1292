"""Drop-in `LatentDiffusion` / `create_latent_diffusion` (reference micro_diffusion/models/model.py:22-405).

Training path (`forward(batch)` -> EDM loss): sigma sampling, noise add, preconditioning, the whole DiT, the
masked-patch loss and the complete backward run as HIP kernels (engine.py); PyTorch supplies the random draws (in
the reference's order: randn[B,1,1,1], randn_like(x), rand[B,T]) and the autograd hook (`loss.backward()`).
Sampling (`edm_sampler_loop` / `generate`) is inference-only glue around the same HIP forward.
"""
from __future__ import annotations

from functools import partial
import warnings
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import dit as model_zoo
from . import hip

DATA_TYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}


def text_encoder_embedding_format(enc: str):
    """(sequence length, embedding dim) of the supported text encoders (reference utils.py:501-513)."""
    if enc in ("stabilityai/stable-diffusion-2-base", "runwayml/stable-diffusion-v1-5", "CompVis/stable-diffusion-v1-4",
               "openclip:hf-hub:apple/DFN5B-CLIP-ViT-H-14-378"):
        return 77, 1024
    if enc == "DeepFloyd/t5-v1_1-xxl":
        return 120, 4096
    raise ValueError(f"Please specify the sequence and embedding size of {enc} encoder")


class DistLoss:
    """Running mean of the per-batch losses (reference utils.py:598-614, without the torchmetrics dependency)."""

    def __init__(self):
        self.loss, self.batches = 0.0, 0

    def update(self, value):
        self.loss = self.loss + value.detach()
        self.batches += 1

    def compute(self):
        return self.loss.float() / self.batches


class _EDMConfig(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class LatentDiffusion(nn.Module):
    def __init__(self, dit: nn.Module, vae, text_encoder, tokenizer, image_key: str = "image", text_key: str = "captions",
                 image_latents_key: str = "image_latents", text_latents_key: str = "caption_latents",
                 precomputed_latents: bool = True, dtype: str = "bfloat16", latent_res: int = 32, p_mean: float = -0.6,
                 p_std: float = 1.2, train_mask_ratio: float = 0.0):
        super().__init__()
        self.dit = dit
        self.vae = vae
        self.image_key, self.text_key = image_key, text_key
        self.image_latents_key, self.text_latents_key = image_latents_key, text_latents_key
        self.precomputed_latents = precomputed_latents
        self.dtype = dtype
        self.latent_res = latent_res
        self.edm_config = _EDMConfig(sigma_min=0.002, sigma_max=80, P_mean=p_mean, P_std=p_std, sigma_data=0.9, num_steps=18,
                                     rho=7, S_churn=0, S_min=0, S_max=float("inf"), S_noise=1)
        self.train_mask_ratio = train_mask_ratio
        self.eval_mask_ratio = 0.0
        assert self.train_mask_ratio >= 0, "Masking ratio must be non-negative!"
        self.randn_like = torch.randn_like
        self.latent_scale = self.vae.config.scaling_factor
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        for frozen in (self.text_encoder, self.vae):
            if isinstance(frozen, nn.Module):
                frozen.requires_grad_(False)

    # ---------------------------------------------------------------------------------------- training step
    def _inputs(self, batch: dict):
        """(latents, conditioning, per-sample caption scale or None) of a batch (model.py:105-135)."""
        if self.precomputed_latents and self.image_latents_key in batch:
            latents = batch[self.image_latents_key]        # already multiplied by the VAE scaling factor
        else:
            with torch.no_grad():
                images = batch[self.image_key]
                latents = self.vae.encode(images.to(DATA_TYPES[self.dtype]))["latent_dist"].sample().data
                latents *= self.latent_scale
        if self.precomputed_latents and self.text_latents_key in batch:
            conditioning = batch[self.text_latents_key]
        else:
            captions = batch[self.text_key]
            captions = captions.view(-1, captions.shape[-1])
            if "attention_mask" in batch:
                conditioning = self.text_encoder.encode(captions, attention_mask=batch["attention_mask"].view(-1, captions.shape[-1]))[0]
            else:
                conditioning = self.text_encoder.encode(captions)[0]
        # Dropped captions (model.py:131-135 multiplies the batch tensor by the 0/1 mask in place; forward() does the same).  For
        # the Trainer's microbatch loop the mask rides along as a per-sample row scale of the first kernel that touches the
        # captions (md_cast_rows_bf16): no torch op on the step path.
        drop = batch.get("drop_caption_mask") if hasattr(batch, "get") else None
        if drop is not None:
            drop = drop.reshape(-1)
            if drop.dtype != torch.float32 or not drop.is_contiguous():
                drop = drop.float().contiguous()
        return latents, conditioning, drop

    def forward(self, batch: dict):
        """(loss, latents, conditioning) as model.py:104-142, including its side effect: dropped captions are zeroed by an
        IN-PLACE multiply of the conditioning tensor (model.py:131-135), so the returned `conditioning` -- and, with precomputed
        latents, the caller's batch['caption_latents'], which is the same tensor -- hold zeros in the dropped rows, exactly what
        a Composer callback / update_metric sees from the reference.  (This is the drop-in surface.  The Trainer's microbatch
        loop, train_microbatch, leaves the batch as the loader produced it and applies the mask as a row scale inside the
        first kernel that reads the captions; the mask is 0 / 1, so applying it on both routes is idempotent.)"""
        latents, conditioning, drop = self._inputs(batch)
        if drop is not None:
            conditioning *= drop.to(conditioning.dtype).view([-1] + [1] * (conditioning.dim() - 1))
        loss = self.edm_loss(latents, conditioning, mask_ratio=self.train_mask_ratio if self.training else self.eval_mask_ratio)
        return (loss, latents, conditioning)

    def _draws(self, x: torch.Tensor, T: int, mask_ratio: float):
        """The three random draws in the reference's order (model.py:182,188, utils.py:390): randn[B,1,1,1], randn_like(x.float()),
        rand[B,T]."""
        B = x.shape[0]
        rnd = torch.randn([B, 1, 1, 1], device=x.device)
        if x.dtype == torch.float32:
            eps = self.randn_like(x)
        elif self.randn_like is torch.randn_like:
            eps = torch.randn_like(x, dtype=torch.float32)
        else:
            eps = self.randn_like(x.float())
        mnoise = torch.rand(B, T, device=x.device) if mask_ratio > 0 else None
        return rnd, eps, mnoise

    def _prep(self, x, y, mask_ratio, _noise):
        dit = self.dit
        dit._ensure_flat()
        dit.refresh_shadow()
        x = x.detach()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        x = x.contiguous()
        B = x.shape[0]
        T = (x.shape[-2] // dit.patch_size) * (x.shape[-1] // dit.patch_size)
        if _noise is None and getattr(self, "_noise_fn", None) is not None:
            _noise = self._noise_fn(B)            # test hook: recorded (rnd_normal, eps, mask_noise) per call
        rnd, eps, mnoise = self._draws(x, T, mask_ratio) if _noise is None else _noise
        if mask_ratio > 0:
            assert dit.training, "Masking is only recommended during training"
        y = y.detach()
        if y.dtype not in (torch.float16, torch.float32):
            y = y.float()
        y = y.contiguous()
        rnd = rnd.reshape(B)
        if rnd.dtype != torch.float32 or not rnd.is_contiguous():
            rnd = rnd.float().contiguous()
        if eps.dtype != torch.float32 or not eps.is_contiguous():
            eps = eps.float().contiguous()
        return x, y, rnd, eps, mnoise

    def edm_loss(self, x: torch.Tensor, y: torch.Tensor, mask_ratio: float = 0, _noise=None, _y_rowscale=None, **kwargs) -> torch.Tensor:
        """model.py:181-210 on the HIP engine.  `_noise=(rnd_normal, eps, mask_noise)` injects the three random
        draws (parity tests); otherwise they are drawn here in the reference's order.  `_y_rowscale` [B] f32: per-sample
        factor on the caption rows (the caption-drop mask)."""
        x, y, rnd, eps, mnoise = self._prep(x, y, mask_ratio, _noise)
        dit = self.dit
        need_grad = torch.is_grad_enabled() and dit._plist[0].requires_grad
        args = (self, dit._grad_anchor, x, y, rnd, eps, mnoise, float(mask_ratio), _y_rowscale)
        if need_grad:
            return _EDMLossFunction.apply(*args)
        return _edm_forward(*args)[0]

    def train_microbatch(self, batch: dict, grad_scale: float = 1.0, loss_accum: Optional[torch.Tensor] = None,
                         accum_weight: float = 0.0, _noise=None) -> torch.Tensor:
        """forward + backward of one microbatch WITHOUT autograd: what `(model(batch)[0] * grad_scale).backward()` does
        (Composer's microbatch loop around model.py:104-142), as one explicit launch sequence — every operation on device
        data is a HIP kernel of libmicrodit_hip (the three random draws are torch's).  Parameter gradients accumulate in the
        flat fp32 buffer (the .grad views); `loss_accum` (1-element f32 tensor) += accum_weight * loss on the device.
        Returns the microbatch loss (device scalar)."""
        latents, conditioning, drop = self._inputs(batch)
        mask_ratio = self.train_mask_ratio if self.training else self.eval_mask_ratio
        x, y, rnd, eps, mnoise = self._prep(latents, conditioning, mask_ratio, _noise)
        dit = self.dit
        loss, tape, dtb = _edm_forward(self, None, x, y, rnd, eps, mnoise, float(mask_ratio), drop,
                                       train=(float(grad_scale), loss_accum, float(accum_weight)))
        dit.attach_grads()
        dit._engine.backward(tape, dtb, on_segment=getattr(dit, "_on_segment", None))
        return loss

    def model_forward_wrapper(self, x, sigma, y, model_forward_fxn, mask_ratio: float, **kwargs) -> dict:
        """EDM preconditioning around an arbitrary forward fn (model.py:144-179); used by the sampler."""
        sd = self.edm_config.sigma_data
        sigma = sigma.to(x.dtype).reshape(-1, 1, 1, 1)
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        c_noise = sigma.log() / 4
        out = model_forward_fxn((c_in * x).to(x.dtype), c_noise.flatten(), y, mask_ratio=mask_ratio, **kwargs)
        out["sample"] = c_skip * x + c_out * out["sample"]
        return out

    # Composer-style hooks (model.py:212-229)
    def loss(self, outputs, batch):
        return outputs[0]

    def eval_forward(self, batch, outputs=None):
        if outputs is not None:
            return outputs
        loss, _, _ = self.forward(batch)
        return loss, None, None

    def get_metrics(self, is_train: bool = False):
        return {"loss": DistLoss()}

    def update_metric(self, batch, outputs, metric):
        metric.update(outputs[0])

    # ---------------------------------------------------------------------------------------- sampling (inference glue)
    @torch.no_grad()
    def edm_sampler_loop(self, x, y, steps: Optional[int] = None, cfg: float = 1.0, fused: Optional[bool] = None, **kwargs):
        """Heun 2nd-order EDM sampler, fp64 state (model.py:231-297).  `fused` (None = whenever possible) selects the loop whose
        per-step arithmetic runs in two fused HIP kernels; False keeps the reference's tensor-op formulation (generic forward
        functions, S_churn > 0)."""
        ec = self.edm_config
        can_fuse = ec.S_churn == 0 and not kwargs and x.is_cuda
        if fused is None:
            fused = can_fuse
        if fused:
            assert can_fuse, "the fused sampler needs S_churn == 0 and no extra forward arguments"
            return self._edm_sampler_fused(x, y, steps, cfg)
        fwd = partial(self.dit.forward, cfg=cfg) if cfg > 1.0 else self.dit.forward
        n = ec.num_steps if steps is None else steps
        idx = torch.arange(n, dtype=torch.float64, device=x.device)
        inv_rho = 1 / ec.rho
        t_steps = (ec.sigma_max ** inv_rho + idx / (n - 1) * (ec.sigma_min ** inv_rho - ec.sigma_max ** inv_rho)) ** ec.rho
        t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
        x_next = x.to(torch.float64) * t_steps[0]
        for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
            x_cur = x_next
            gamma = min(ec.S_churn / n, np.sqrt(2) - 1) if ec.S_min <= t_cur <= ec.S_max else 0
            t_hat = torch.as_tensor(t_cur + gamma * t_cur)
            x_hat = x_cur + (t_hat ** 2 - t_cur ** 2).sqrt() * ec.S_noise * self.randn_like(x_cur)
            den = self.model_forward_wrapper(x_hat.to(torch.float32), t_hat.to(torch.float32), y, fwd, mask_ratio=0, **kwargs)["sample"].to(torch.float64)
            d_cur = (x_hat - den) / t_hat
            x_next = x_hat + (t_next - t_hat) * d_cur
            if i < n - 1:
                den = self.model_forward_wrapper(x_next.to(torch.float32), t_next.to(torch.float32), y, fwd, mask_ratio=0, **kwargs)["sample"].to(torch.float64)
                d_prime = (x_next - den) / t_next
                x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
        return x_next.to(torch.float32)

    @torch.no_grad()
    def _edm_sampler_fused(self, x, y, steps: Optional[int], cfg: float):
        """The same Heun loop (S_churn = 0, the reference's setting: x_hat = x_cur) with everything around the network evaluations in
        two fused HIP kernels: md_edm_sampler_input (c_in scaling + guidance batch doubling) and md_edm_heun_update (guidance
        combine + preconditioning + fp64 Euler / Heun update)."""
        ec, L, st = self.edm_config, hip.lib(), torch.cuda.current_stream().cuda_stream
        n = ec.num_steps if steps is None else steps
        idx = torch.arange(n, dtype=torch.float64)
        inv_rho = 1 / ec.rho
        t_steps = (ec.sigma_max ** inv_rho + idx / (n - 1) * (ec.sigma_min ** inv_rho - ec.sigma_max ** inv_rho)) ** ec.rho
        t_steps = torch.cat([t_steps, torch.zeros(1, dtype=torch.float64)]).tolist()
        guided = cfg > 1.0
        B, numel = x.shape[0], x.numel()
        x_cur = (x.to(torch.float64) * t_steps[0]).contiguous()
        x_nxt, d_cur = torch.empty_like(x_cur), torch.empty_like(x_cur)
        net_in = torch.empty((2 * B if guided else B,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        y2 = torch.cat([y, torch.zeros_like(y)], 0) if guided else y

        def network(xs, sigma):
            hip.check(L.md_edm_sampler_input(xs.data_ptr(), net_in.data_ptr(), numel, float(sigma), ec.sigma_data, 1 if guided else 0, st),
                      "md_edm_sampler_input")
            t = torch.full((1,), float(np.log(np.float32(sigma)) / 4), device=x.device, dtype=torch.float32)
            return self.dit.forward_without_cfg(net_in, t, y2, 0)["sample"].contiguous()
        for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
            F = network(x_cur, t_cur)
            hip.check(L.md_edm_heun_update(x_cur.data_ptr(), x_cur.data_ptr(), F.data_ptr(), d_cur.data_ptr(), x_nxt.data_ptr(), numel,
                                           float(cfg), 1 if guided else 0, t_cur, t_cur, t_next, ec.sigma_data, 0, st), "md_edm_heun_update")
            if i < n - 1:
                F = network(x_nxt, t_next)
                hip.check(L.md_edm_heun_update(x_cur.data_ptr(), x_nxt.data_ptr(), F.data_ptr(), d_cur.data_ptr(), x_nxt.data_ptr(), numel,
                                               float(cfg), 1 if guided else 0, t_next, t_cur, t_next, ec.sigma_data, 1, st), "md_edm_heun_update")
            x_cur, x_nxt = x_nxt, x_cur
        return x_cur.to(torch.float32)

    @torch.no_grad()
    def generate(self, prompt: Optional[list] = None, tokenized_prompts=None, attention_mask=None, guidance_scale: float = 5.0,
                 num_inference_steps: int = 30, seed: Optional[int] = None, return_only_latents: bool = False, **kwargs):
        """tokenise -> text encoder -> EDM sampler on the HIP DiT -> VAE decode (model.py:299-353)."""
        assert prompt or tokenized_prompts is not None, "Must provide either prompt or tokenized prompts"
        device = next(self.dit.parameters()).device
        gen = torch.Generator(device=device)
        if seed:
            gen = gen.manual_seed(seed)
        if tokenized_prompts is None:
            out = self.tokenizer.tokenize(prompt)
            tokenized_prompts = out["input_ids"]
            attention_mask = out.get("attention_mask")
        emb = self.text_encoder.encode(tokenized_prompts.to(device),
                                       attention_mask=attention_mask.to(device) if attention_mask is not None else None)[0]
        latents = torch.randn((len(emb), self.dit.in_channels, self.latent_res, self.latent_res), device=device, generator=gen)
        latents = self.edm_sampler_loop(latents, emb, num_inference_steps, cfg=guidance_scale)
        if return_only_latents:
            return latents
        image = self.vae.decode((latents / self.latent_scale).to(DATA_TYPES[self.dtype])).sample
        return (image / 2 + 0.5).clamp(0, 1).float().detach()


def _edm_forward(model: LatentDiffusion, anchor, x, y, rnd, eps, mnoise, mask_ratio, y_rowscale=None, train=None,
                 record_tape: bool = False):
    """Forward half of the fused training step.  Returns (loss scalar tensor, tape, dtok).
    train = (grad_scale, loss_accum, accum_weight): the Trainer's autograd-free form — the tape goes to the engine's
    fixed-address arena and dtok comes back as bf16, already multiplied by grad_scale (md_edm_loss_train).  Otherwise dtok is
    fp32 and unscaled (the autograd node scales it by the upstream gradient)."""
    dit = model.dit
    eng = dit._engine
    L = hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    ec = model.edm_config
    B, C, H, W = x.shape
    dev = x.device
    xn = torch.empty(x.shape, device=dev, dtype=torch.float32)
    sigma, cin, cnoise = (torch.empty(B, device=dev) for _ in range(3))
    if x.dtype == torch.float16:
        x0 = torch.empty(x.shape, device=dev, dtype=torch.float32)
        hip.check(L.md_edm_prepare_f16(x.data_ptr(), eps.data_ptr(), rnd.data_ptr(), xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(),
                                       cin.data_ptr(), cnoise.data_ptr(), B, C * H * W, ec.P_mean, ec.P_std, ec.sigma_data, st),
                  "md_edm_prepare_f16")
    else:
        x0 = x
        hip.check(L.md_edm_prepare(x.data_ptr(), eps.data_ptr(), rnd.data_ptr(), xn.data_ptr(), sigma.data_ptr(), cin.data_ptr(),
                                   cnoise.data_ptr(), B, C * H * W, ec.P_mean, ec.P_std, ec.sigma_data, st), "md_edm_prepare")
    tape = eng.forward(xn, cnoise, y, mask_ratio=mask_ratio, mask_noise=mnoise, in_scale=cin, y_rowscale=y_rowscale,
                       record_tape=record_tape or train is not None, arena=train is not None)
    lps = torch.empty(B, device=dev)
    loss = torch.empty(1, device=dev)
    keep = None if tape.keep_rows is None else tape.keep_rows.data_ptr()
    if train is not None:
        gscale, accum, aw = train
        dtok = torch.empty(B * tape.Tk, dit.config.patch_vec, device=dev, dtype=torch.bfloat16)
        hip.check(L.md_edm_loss_train(tape.out_tok.data_ptr(), keep, xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(), lps.data_ptr(),
                                      loss.data_ptr(), dtok.data_ptr(), gscale, None if accum is None else accum.data_ptr(), aw, B,
                                      tape.Tk, C, H, W, dit.patch_size, ec.sigma_data, st), "md_edm_loss_train")
    else:
        dtok = torch.empty(B * tape.Tk, dit.config.patch_vec, device=dev) if record_tape else None
        hip.check(L.md_edm_loss(tape.out_tok.data_ptr(), keep, xn.data_ptr(), x0.data_ptr(), sigma.data_ptr(), lps.data_ptr(),
                                loss.data_ptr(), None if dtok is None else dtok.data_ptr(), B, tape.Tk, C, H, W, dit.patch_size,
                                ec.sigma_data, st), "md_edm_loss")
    tape.loss_per_sample = lps
    return loss.reshape(()), tape, dtok


class _EDMLossFunction(torch.autograd.Function):
    """loss = EDM(x, y) as one autograd node (the drop-in `loss.backward()` surface); backward runs the engine's hand-written
    backward and accumulates the parameter gradients straight into the flat fp32 grad buffer (the .grad views)."""

    @staticmethod
    def forward(ctx, model, anchor, x, y, rnd, eps, mnoise, mask_ratio, y_rowscale):
        loss, tape, dtok = _edm_forward(model, anchor, x, y, rnd, eps, mnoise, mask_ratio, y_rowscale, record_tape=True)
        ctx.model, ctx.tape, ctx.dtok = model, tape, dtok
        return loss

    @staticmethod
    def backward(ctx, gloss):
        model, tape, dtok = ctx.model, ctx.tape, ctx.dtok
        dit = model.dit
        dit.attach_grads()
        g = gloss.detach().to(torch.float32).contiguous()
        dtb = torch.empty(dtok.shape, device=dtok.device, dtype=torch.bfloat16)
        hip.check(hip.lib().md_cast_f32_bf16(dtok.data_ptr(), dtb.data_ptr(), dtok.numel(), g.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "md_cast_f32_bf16")
        dit._engine.backward(tape, dtb, on_segment=getattr(dit, "_on_segment", None))
        ctx.tape = ctx.dtok = None
        return None, torch.zeros_like(dit._grad_anchor), None, None, None, None, None, None, None


class _FrozenStub(nn.Module):
    """Placeholder for the frozen SDXL-VAE / text encoder when their libraries (diffusers / open_clip) are absent:
    enough for training on precomputed latents (train.py asserts precomputed_latents), loud on any real use."""

    def __init__(self, what: str, scaling_factor: float = 0.13025):
        super().__init__()
        self.what = what
        self.config = SimpleNamespace(scaling_factor=scaling_factor)

    def _fail(self, *a, **k):
        raise RuntimeError(f"{self.what} is not available in this environment (library not installed); "
                           "only precomputed-latent training is possible")

    encode = decode = forward = tokenize = _fail


def create_latent_diffusion(vae_name: str = "stabilityai/stable-diffusion-xl-base-1.0",
                            text_encoder_name: str = "openclip:hf-hub:apple/DFN5B-CLIP-ViT-H-14-378",
                            dit_arch: str = "MicroDiT_XL_2", latent_res: int = 32, in_channels: int = 4,
                            pos_interp_scale: float = 1.0, dtype: str = "bfloat16", precomputed_latents: bool = True,
                            p_mean: float = -0.6, p_std: float = 1.2, train_mask_ratio: float = 0.0) -> LatentDiffusion:
    """Same signature and behaviour as the reference factory (model.py:356-405); the DiT comes from this package's
    zoo; the frozen VAE / text encoder are loaded through diffusers / open_clip when those are installed."""
    s, d = text_encoder_embedding_format(text_encoder_name)
    dit = getattr(model_zoo, dit_arch)(input_size=latent_res, caption_channels=d, pos_interp_scale=pos_interp_scale,
                                       in_channels=in_channels)
    vae = text_encoder = tokenizer = None
    try:
        from diffusers import AutoencoderKL  # type: ignore
    except ImportError:
        warnings.warn(f"diffusers is not installed: VAE '{vae_name}' replaced by a stub (training on precomputed latents only)")
        vae = _FrozenStub(f"VAE '{vae_name}'")
    else:       # a real load error (bad name, no network) must surface, not turn into a stub with a made-up scaling factor
        vae = AutoencoderKL.from_pretrained(vae_name, subfolder=None if vae_name == "ostris/vae-kl-f8-d16" else "vae",
                                            torch_dtype=DATA_TYPES[dtype])
    # The reference's UniversalTextEncoder / UniversalTokenizer (utils.py:429-598: open_clip / T5 / CLIP wrappers around
    # frozen third-party models) are outside the training hot path and are NOT built here (out of
    # scope): generate(prompt=...) needs caller-supplied embeddings; training reads precomputed caption latents.
    text_encoder = _FrozenStub(f"text encoder '{text_encoder_name}' (wrapper not built: pass precomputed caption embeddings)")
    tokenizer = _FrozenStub(f"tokenizer '{text_encoder_name}' (wrapper not built)")
    return LatentDiffusion(dit=dit, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer,
                           precomputed_latents=precomputed_latents, dtype=dtype, latent_res=latent_res, p_mean=p_mean,
                           p_std=p_std, train_mask_ratio=train_mask_ratio)
