"""Parity of the complete HIP training path (LatentDiffusion.forward -> EDM loss -> backward) against the CPU oracle
(oracle/microdit_ref.py, itself pinned to the reference by tests/golden) on identical weights, latents, captions
and noise.  Tolerances: the HIP path is bf16 storage / fp32 accumulate and is
compared with the fp32 oracle: network output rel-RMS <= 3 %, loss within 1 %, mask bit-exact; gradients: global
direction (cosine) and norm, per-tensor rel-RMS reported and bounded loosely (bf16 activations + top-k flips)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import microdit_ref as orc

pytestmark = pytest.mark.gpu

CASES = [("tiny_mask75", orc.tiny_config, 4, 11, 0.75, -0.6, 1.2, 77),
         ("tiny_mask0", orc.tiny_config, 2, 12, 0.0, -0.6, 1.2, 77),
         ("micro_mask50", orc.micro_config, 3, 13, 0.5, 0.0, 0.6, 20)]


def _rel_rms(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)))


def build_product(cfg, sd, p_mean, p_std, ratio):
    from micro_diffusion_amd import dit as mdit
    from micro_diffusion_amd.model import LatentDiffusion, _FrozenStub
    d = mdit.DiT(**cfg.__dict__)
    missing = d.load_state_dict(sd, strict=True)
    d = d.to("cuda")
    m = LatentDiffusion(d, _FrozenStub("vae"), _FrozenStub("te"), _FrozenStub("tok"), p_mean=p_mean, p_std=p_std,
                        train_mask_ratio=ratio)
    m.train()
    return m


@pytest.mark.parametrize("tag,cfgf,B,seed,ratio,pm,ps,cap", CASES)
def test_train_step_parity(hip, tag, cfgf, B, seed, ratio, pm, ps, cap):
    cfg = cfgf()
    sd = orc.synth_state_dict(cfg, seed)
    batch, rnd, epsn, mnoise = orc.synth_batch(cfg, B, seed + 1, cap_len=cap)
    # ---- oracle (CPU fp32)
    osd = {k: v.clone().requires_grad_(k not in ("pos_embed", "mask_token")) for k, v in sd.items()}
    oloss = orc.latent_diffusion_forward(osd, cfg, batch, rnd, epsn, mnoise, ratio, pm, ps)
    oloss.backward()
    sigma = (rnd * ps + pm).exp()
    xin = (batch["image_latents"].float() + epsn * sigma) / (0.9 ** 2 + sigma ** 2).sqrt()
    cond = batch["caption_latents"].float() * batch["drop_caption_mask"].view(-1, 1, 1, 1)
    with torch.no_grad():
        osample, omask = orc.dit_forward(osd, cfg, xin, (sigma.log() / 4).flatten(), cond, ratio, mnoise)
    # ---- golden cross-check of the oracle numbers used here (loss recorded from the reference itself)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
    assert abs(oloss.item() - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    # ---- HIP path
    model = build_product(cfg, sd, pm, ps, ratio)
    gb = {k: v.cuda() for k, v in batch.items()}
    noise = (rnd.cuda(), epsn.cuda(), mnoise.cuda() if ratio > 0 else None)
    lat = gb["image_latents"]
    condg = gb["caption_latents"] * gb["drop_caption_mask"].view(-1, 1, 1, 1).half()
    loss = model.edm_loss(lat, condg, mask_ratio=ratio, _noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    # raw network output through the plain DiT API
    with torch.no_grad():
        out = model.dit(xin.cuda(), (sigma.log() / 4).flatten().cuda(), cond.cuda(), mask_ratio=ratio,
                        mask_noise=mnoise.cuda() if ratio > 0 else None)
    torch.cuda.synchronize()
    rep = {"case": tag, "loss_hip": loss.item(), "loss_oracle": oloss.item()}
    rr = _rel_rms(out["sample"].cpu(), osample)
    rep["sample_rel_rms"] = rr
    grads = {k: p.grad.detach().cpu() for k, p in model.dit.named_parameters()}
    per = {k: _rel_rms(grads[k], osd[k].grad) for k in grads}
    gn_h = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    gn_o = float(torch.sqrt(sum((osd[k].grad.double() ** 2).sum() for k in grads)))
    dot = float(sum((grads[k].double() * osd[k].grad.double()).sum() for k in grads))
    rep.update(gnorm_hip=gn_h, gnorm_oracle=gn_o, cosine=dot / (gn_h * gn_o), worst=sorted(per.items(), key=lambda kv: -kv[1])[:12])
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/engine_parity_{tag}.json", "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps(rep, indent=1))
    if ratio > 0:
        assert torch.equal(out["mask"].cpu(), omask), "mask selection must be bit-exact"
    assert rr <= 0.03, f"network output rel-RMS {rr}"
    assert abs(loss.item() - oloss.item()) <= 0.01 * abs(oloss.item()), (loss.item(), oloss.item())
    assert rep["cosine"] >= 0.99, rep["cosine"]
    assert abs(gn_h - gn_o) <= 0.03 * gn_o
    bad = {k: v for k, v in per.items() if v > 0.15}
    assert not bad, bad


@pytest.mark.parametrize("cfgf,B,seed,ratio,cap", [(orc.tiny_config, 4, 21, 0.75, 77), (orc.micro_config, 3, 22, 0.5, 20)])
def test_head_major_qk_equals_packed(hip, cfgf, B, seed, ratio, cap):
    """DiTEngine.qk_head_major (q / k as [B, H, S, hd] through md_qkln_fwd_hm / md_qkln_bwd_hm, off by default) is a LAYOUT: the same
    kernels do the same arithmetic on the same values, so the loss must be bit-identical and every gradient equal to 1e-2 rel-RMS to the packed-row step
    (self-attention, cross-attention with 77 / 20 caption tokens, the caption attention block)."""
    cfg = cfgf()
    sd = orc.synth_state_dict(cfg, seed)
    batch, rnd, epsn, mnoise = orc.synth_batch(cfg, B, seed + 1, cap_len=cap)
    gb = {k: v.cuda() for k, v in batch.items()}
    noise = (rnd.cuda(), epsn.cuda(), mnoise.cuda())
    condg = gb["caption_latents"] * gb["drop_caption_mask"].view(-1, 1, 1, 1).half()
    got = {}
    for hm in (False, True):
        model = build_product(cfg, sd, -0.6, 1.2, ratio)
        model.dit.engine.qk_head_major = hm
        loss = model.edm_loss(gb["image_latents"], condg, mask_ratio=ratio, _noise=noise)
        loss.backward()
        torch.cuda.synchronize()
        got[hm] = (loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.dit.named_parameters()})
    assert torch.equal(got[True][0], got[False][0]), (got[True][0].item(), got[False][0].item())
    # (a handful of reductions -- adaLN gate / shift sums -- use fp32 atomics: their order, not the layout, moves the last bits)
    diff = {k: _rel_rms(got[True][1][k], got[False][1][k]) for k in got[False][1]}
    # (run-to-run spread of the atomics alone: 5e-4 typical, 2.9e-3 seen once on the pooled-caption MLP's fc1 -- every adaLN sum of the
    # network feeds it; a layout mistake moves a tensor by O(1), so the bound sits an order of magnitude above the spread)
    bad = {k: v for k, v in diff.items() if v > 1e-2}
    assert not bad, bad


def _res512_cfg():
    """configs/res_512_*.yaml geometry on the Tiny widths: 64x64 latents (T = 1024 tokens), pos_interp_scale = 2."""
    return orc.tiny512_config()


@pytest.mark.parametrize("ratio", [0.75, 0.0])
def test_res512_stage_parity(hip, ratio):
    """Stage 3 / 4 shapes (BASELINE.json configs[3..4]): T = 1024 mixer tokens, 256 (mask 0.75) or 1024 (mask 0) backbone
    tokens, P_mean 0 / P_std 0.6 — same tolerances as the 256-res cases."""
    cfg = _res512_cfg()
    seed, B, pm, ps = 31, 2, 0.0, 0.6
    sd = orc.synth_state_dict(cfg, seed)
    batch, rnd, epsn, mnoise = orc.synth_batch(cfg, B, seed + 1)
    osd = {k: v.clone().requires_grad_(k not in ("pos_embed", "mask_token")) for k, v in sd.items()}
    oloss = orc.latent_diffusion_forward(osd, cfg, batch, rnd, epsn, mnoise, ratio, pm, ps)
    oloss.backward()
    model = build_product(cfg, sd, pm, ps, ratio)
    cond = (batch["caption_latents"] * batch["drop_caption_mask"].view(-1, 1, 1, 1).half()).cuda()
    loss = model.edm_loss(batch["image_latents"].cuda(), cond, mask_ratio=ratio,
                          _noise=(rnd.cuda(), epsn.cuda(), mnoise.cuda() if ratio > 0 else None))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - oloss.item()) <= 0.01 * abs(oloss.item()), (loss.item(), oloss.item())
    grads = {k: p.grad.detach().cpu() for k, p in model.dit.named_parameters()}
    gn_h = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    gn_o = float(torch.sqrt(sum((osd[k].grad.double() ** 2).sum() for k in grads)))
    dot = float(sum((grads[k].double() * osd[k].grad.double()).sum() for k in grads))
    assert dot / (gn_h * gn_o) >= 0.99 and abs(gn_h - gn_o) <= 0.03 * gn_o, (dot / (gn_h * gn_o), gn_h, gn_o)
    if ratio > 0:       # mask of the 1024-token row: bit-exact
        with torch.no_grad():
            out = model.dit(batch["image_latents"].float().cuda(), torch.zeros(B).cuda(), cond, mask_ratio=ratio,
                            mask_noise=mnoise.cuda())
        assert torch.equal(out["mask"].cpu(), orc.get_mask(mnoise, ratio)["mask"])


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] geometry: MicroDiT_XL_2 (dit.py:671-709) — head_dim 64, 8..16 heads varying per layer, FFN
# hidden 512..3840, 8 experts, mixer 6 x 768 -> backbone 28 x 1024, 1.165 G parameters.  Round 1 only checked the engine
# at Tiny widths (head_dim 32, dim <= 256); this is the same train-step parity at the widths the benchmark runs, once with
# the library's kernel choice and once with the persistent pp256 GEMM forced wherever it accepts the problem, so the
# per-layer width plan, the fused [w1; w2] launches, split-K factors and 8-expert grouped launches are all exercised
# against the oracle at XL/2 geometry.  (The oracle itself is pinned to the reference at XL/2 by tests/golden/xl2_mask75.npz.)
# ---------------------------------------------------------------------------------------------------------------------
_XL2 = {}
# tag -> (oracle config, batch, seed, mask ratio, P_mean, P_std): BASELINE.json configs[1], [3], [4] at XL/2 widths
XL2_CASES = {
    "xl2_mask75": (lambda: orc.xl2_config(), 2, 41, 0.75, -0.6, 1.2),                                   # res_256_pretrain.yaml
    "xl2_mask0": (lambda: orc.xl2_config(), 2, 43, 0.0, -0.6, 1.2),                                     # res_256_finetune.yaml
    "xl2_res512_mask75": (lambda: orc.xl2_config(input_size=64, pos_interp_scale=2.0), 1, 45, 0.75, 0.0, 0.6),   # res_512_pretrain.yaml
}


def _xl2_oracle(tag):
    if tag not in _XL2:
        cfgf, B, seed, ratio, pm, ps = XL2_CASES[tag]
        cfg = cfgf()
        sd = orc.synth_state_dict(cfg, seed)
        batch, rnd, epsn, mnoise = orc.synth_batch(cfg, B, seed + 1)
        osd = {k: v.clone().requires_grad_(k not in ("pos_embed", "mask_token")) for k, v in sd.items()}
        taps = {}
        oloss = orc.latent_diffusion_forward(osd, cfg, batch, rnd, epsn, mnoise, ratio, pm, ps, taps=taps)
        oloss.backward()
        _XL2[tag] = dict(cfg=cfg, sd=sd, batch=batch, noise=(rnd, epsn, mnoise), loss=float(oloss.detach()),
                         grads={k: v.grad for k, v in osd.items() if v.grad is not None}, ratio=ratio, pm=pm, ps=ps,
                         taps={k: v.detach() for k, v in taps.items()})
    return _XL2[tag]


def _gain(a, b):
    """Least-squares scale of a onto b: <a, b> / <b, b> (1 = no systematic shrink / growth of the product's activations)."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (b * b).sum())


def _block_drift(tape, taps, cfg, fn=_rel_rms):
    """rel-RMS (or gain) of the residual stream after every block (product tape, bf16) against the oracle's (fp32): localises
    where a loss difference is made.  tape.mixer[i].x / tape.blocks[i].x are the block INPUTS; the last backbone output is tape.xlast."""
    mixer_specs, block_specs = orc.block_specs(cfg)
    out = {}
    outs_m = [t.x for t in tape.mixer[1:]]
    for spec, x in zip(mixer_specs[:-1], outs_m):
        out[spec.prefix] = fn(x.float().cpu().view(-1), taps["out::" + spec.prefix].reshape(-1))
    outs_b = [t.x for t in tape.blocks[1:]] + [tape.xlast]
    for spec, x in zip(block_specs, outs_b):
        out[spec.prefix] = fn(x.float().cpu().view(-1), taps["out::" + spec.prefix].reshape(-1))
    return out


# every geometry with the library's kernel choice; the persistent kernel forced wherever it accepts the problem on the headline
# geometry (the AUTO rules already send most launches of the other two there; each XL/2 case costs about a minute of host time)
XL2_RUNS = [(tag, "auto") for tag in XL2_CASES] + [("xl2_mask75", "pp256")]


@pytest.mark.parametrize("tag,prefer", XL2_RUNS)
def test_xl2_train_step_parity(hip, tag, prefer):
    """Whole train step at XL/2 widths for the three stage geometries the benchmark times (BASELINE.json configs[1], [3], [4];
    goldens recorded from the unmodified reference: oracle/gen_golden.py xl2 / xl2_mask0 / xl2_res512)."""
    o = _xl2_oracle(tag)
    cfg, sd, batch = o["cfg"], o["sd"], o["batch"]
    rnd, epsn, mnoise = o["noise"]
    ratio = o["ratio"]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
    assert abs(o["loss"] - float(z["loss"])) <= 2e-5 * abs(float(z["loss"])), "oracle differs from the reference at XL/2"
    model = build_product(cfg, sd, o["pm"], o["ps"], ratio)
    eng = model.dit.engine
    eng.gemm_prefer = hip.GEMM_VARIANT_NAMES[prefer]
    eng.gemm_log = []
    eng.keep_last_tape = True
    cond = (batch["caption_latents"] * batch["drop_caption_mask"].view(-1, 1, 1, 1).half()).cuda()
    noise = (rnd.cuda(), epsn.cuda(), mnoise.cuda() if ratio > 0 else None)
    loss = model.edm_loss(batch["image_latents"].cuda(), cond, mask_ratio=ratio, _noise=noise)
    loss.backward()
    torch.cuda.synchronize()
    drift = _block_drift(eng.last_tape, o["taps"], cfg)
    used = {v for v, *_ in eng.gemm_log}
    if prefer == "pp256":
        n_pp = sum(1 for v, *_ in eng.gemm_log if v == hip.GEMM_PP256)
        assert n_pp >= 0.5 * len(eng.gemm_log), f"pp256 ran only {n_pp} of {len(eng.gemm_log)} GEMM launches"
    grads = {k: p.grad.detach().cpu() for k, p in model.dit.named_parameters()}
    og = o["grads"]
    per = {k: _rel_rms(grads[k], og[k]) for k in grads}
    gn_h = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    gn_o = float(torch.sqrt(sum((og[k].double() ** 2).sum() for k in grads)))
    dot = float(sum((grads[k].double() * og[k].double()).sum() for k in grads))
    # ---- the same forward with the ORACLE's expert-choice indices injected into every routed layer: what is left of the loss /
    # residual-stream difference is bf16 arithmetic; what disappeared was top-k slots ranked differently by bf16 gate logits
    # -- forward AND backward: with the routing equalised every tensor -- gate weights, the LayerNorm that feeds the router and
    # the expert weights of the 20 routed layers included -- must meet the 15 % per-tensor bound of the small configs
    eng.route_override = {k[len("route::"):]: v for k, v in o["taps"].items() if k.startswith("route::")}
    for prm in model.dit.parameters():
        prm.grad = None                                  # attach_grads() re-creates the views over a zeroed accumulator
    loss_r = model.edm_loss(batch["image_latents"].cuda(), cond, mask_ratio=ratio, _noise=noise)
    loss_r.backward()
    torch.cuda.synchronize()
    grads_r = {k: p.grad.detach().cpu() for k, p in model.dit.named_parameters()}
    per_r = {k: _rel_rms(grads_r[k], og[k]) for k in grads_r}
    gn_r = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads_r.values())))
    dot_r = float(sum((grads_r[k].double() * og[k].double()).sum() for k in grads_r))
    loss_r = loss_r.detach()
    drift_r = _block_drift(eng.last_tape, o["taps"], cfg)
    gain_r = _block_drift(eng.last_tape, o["taps"], cfg, fn=_gain)
    F_hip = eng.sample_image(eng.last_tape).float().cpu()
    F_rep = {"rel_rms": _rel_rms(F_hip, o["taps"]["F"]), "gain": _gain(F_hip, o["taps"]["F"]),
             "per_sample_gain": [_gain(F_hip[i], o["taps"]["F"][i]) for i in range(F_hip.shape[0])]}
    # the final layer (utils.py:236-240) op by op against the oracle's own intermediates: shift / scale = adaLN(gelu(c)),
    # modulated LayerNorm of the last residual stream, the p*p*C projection (kept tokens, before unmask / unpatchify)
    tp, tap, sdf = eng.last_tape, o["taps"], {k: v.float() for k, v in sd.items() if k.startswith("final_layer.")}
    D = cfg.dim
    gco = torch.nn.functional.gelu(tap["c"], approximate="tanh")
    fmod_o = gco @ sdf["final_layer.adaLN_modulation.1.weight"].t() + sdf["final_layer.adaLN_modulation.1.bias"]
    hb = tap["backbone_out"]
    xf_o = torch.nn.functional.layer_norm(hb, (D,), sdf["final_layer.norm_final.weight"], None, cfg.norm_eps)
    xf_o = xf_o * (1 + fmod_o[:, D:].unsqueeze(1)) + fmod_o[:, :D].unsqueeze(1)
    tok_o = xf_o @ sdf["final_layer.linear.weight"].t() + sdf["final_layer.linear.bias"]
    fm_h = tp.fmod.float().cpu()
    F_rep["final_layer"] = {
        "c": {"gain": _gain(tp.c.float().cpu(), tap["c"]), "rel_rms": _rel_rms(tp.c.float().cpu(), tap["c"])},
        "shift": {"gain": _gain(fm_h[:, :D], fmod_o[:, :D]), "rel_rms": _rel_rms(fm_h[:, :D], fmod_o[:, :D])},
        "scale": {"gain": _gain(fm_h[:, D:], fmod_o[:, D:]), "rel_rms": _rel_rms(fm_h[:, D:], fmod_o[:, D:])},
        "modulated_ln": {"gain": _gain(tp.xf.float().cpu().view(-1), xf_o.reshape(-1)), "rel_rms": _rel_rms(tp.xf.float().cpu().view(-1), xf_o.reshape(-1))},
        "tokens": {"gain": _gain(tp.out_tok.float().cpu().view(-1), tok_o.reshape(-1)), "rel_rms": _rel_rms(tp.out_tok.float().cpu().view(-1), tok_o.reshape(-1))},
        # the same projected error split into its token-constant part (per-sample mean over the kept tokens: shift / bias path and
        # the mean residual error, 16 numbers per sample -> few degrees of freedom) and the token-varying rest
        "tokens_token_mean_part": {"gain": _gain(tp.out_tok.float().cpu().view(tok_o.shape).mean(1), tok_o.mean(1)),
                                   "share_of_signal_power": float(tok_o.mean(1, keepdim=True).expand_as(tok_o).pow(2).sum() / tok_o.pow(2).sum())},
        "tokens_centered_part": {"gain": _gain(tp.out_tok.float().cpu().view(tok_o.shape) - tp.out_tok.float().cpu().view(tok_o.shape).mean(1, keepdim=True),
                                               tok_o - tok_o.mean(1, keepdim=True))},
        "tokens_from_oracle_xf_bf16_weights": {"gain": _gain((xf_o.bfloat16().float() @ sdf["final_layer.linear.weight"].bfloat16().float().t()
                                                                + sdf["final_layer.linear.bias"]).reshape(-1), tok_o.reshape(-1))},
        "rms": {"backbone_out": float(hb.pow(2).mean().sqrt()), "scale": float(fmod_o[:, D:].pow(2).mean().sqrt()),
                "shift": float(fmod_o[:, :D].pow(2).mean().sqrt()), "tokens": float(tok_o.pow(2).mean().sqrt())}}
    eng.route_override = None
    mixer_keys = [k for k in drift if k.startswith("patch_mixer")]
    every4 = mixer_keys[-1:] + [k for k in drift if k.startswith("blocks.") and int(k.split(".")[1]) % 4 == 3]
    rep = {"case": f"{tag}_{prefer}", "loss_hip": loss.item(), "loss_oracle": o["loss"], "loss_rel_diff": (loss.item() - o["loss"]) / o["loss"],
           "loss_hip_oracle_routing": loss_r.item(), "loss_rel_diff_oracle_routing": (loss_r.item() - o["loss"]) / o["loss"],
           "gnorm_hip": gn_h, "gnorm_oracle": gn_o, "gnorm_hip_oracle_routing": gn_r,
           "cosine": dot / (gn_h * gn_o), "cosine_oracle_routing": dot_r / (gn_r * gn_o),
           "worst_oracle_routing": sorted(per_r.items(), key=lambda kv: -kv[1])[:12], "gemm_launches": len(eng.gemm_log), "variants_requested": sorted(used),
           "network_output_F_oracle_routing": F_rep,
           "residual_stream_gain_oracle_routing": {k: gain_r[k] for k in every4},
           "residual_stream_rel_rms": {k: drift[k] for k in every4},
           "residual_stream_rel_rms_oracle_routing": {k: drift_r[k] for k in every4},
           "residual_stream_rel_rms_all": drift, "residual_stream_rel_rms_all_oracle_routing": drift_r,
           "worst": sorted(per.items(), key=lambda kv: -kv[1])[:12]}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/engine_parity_{tag}_{prefer}.json", "w") as fh:
        json.dump(rep, fh, indent=1)
    print(json.dumps({k: v for k, v in rep.items() if not k.startswith("residual_stream_rel_rms_all")}, indent=1))
    assert abs(loss.item() - o["loss"]) <= 0.01 * abs(o["loss"]), (loss.item(), o["loss"])
    assert abs(loss_r.item() - o["loss"]) <= 0.01 * abs(o["loss"]), (loss_r.item(), o["loss"])
    assert rep["cosine"] >= 0.99, rep["cosine"]
    assert abs(gn_h - gn_o) <= 0.03 * gn_o
    assert max(drift_r.values()) <= 0.05, max(drift_r.items(), key=lambda kv: kv[1])     # bf16 residual stream, routing equalised
    # Per-tensor bound: 15 % as for the small configs, except for the tensors whose gradient passes through the expert-choice
    # top-k of a 28-layer model at batch 2 (gate weights, the LayerNorm that feeds the router, expert weights): with 128 tokens
    # per layer ONE slot that the bf16 gate logits rank differently from the fp32 oracle moves such a gradient by 10-30 %
    # (the reference's own bf16-vs-fp32 run does the same).  Their arithmetic at this geometry is bounded
    # separately with the oracle's routing injected (tests/test_kernels_gpu.py::test_moe_layer_with_oracle_routing, 1 %).
    mixer_specs, block_specs = orc.block_specs(cfg)
    moe_blocks = {s.prefix for s in mixer_specs + block_specs if s.moe}

    def routed(name):
        pre = ".".join(name.split(".")[:2])
        return pre in moe_blocks and (".mlp." in name or ".norm3." in name)
    bad = {k: v for k, v in per.items() if v > (0.40 if routed(k) else 0.15)}      # the 40 % allowance: own routing ONLY
    assert not bad, bad
    # ---- with the oracle's routing: the network output meets the 3 % bar and EVERY gradient tensor 15 %
    assert F_rep["rel_rms"] <= 0.03, F_rep["rel_rms"]
    assert rep["cosine_oracle_routing"] >= 0.99 and abs(gn_r - gn_o) <= 0.03 * gn_o, (rep["cosine_oracle_routing"], gn_r, gn_o)
    bad_r = {k: v for k, v in per_r.items() if v > 0.15}
    assert not bad_r, bad_r
