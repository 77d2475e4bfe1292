"""Hydra-free loader for the reference's YAML stage configs (reference configs/*.yaml, train.py:14-22):
`--config-path DIR --config-name FILE key.sub=value ...`, `${a.b}` interpolation, `_target_` instantiation with the
reference's class paths mapped onto this package's MI355X-native equivalents."""
from __future__ import annotations

import importlib
import os
import re
from typing import Any, Dict, List

import yaml

# reference `_target_` -> native implementation ("module:attr"); None = accepted and ignored (host-side observability
# that has no counterpart on the hot path: loggers / monitors; see DESIGN.md "out of scope").
TARGETS = {
    "micro_diffusion.models.model.create_latent_diffusion": "micro_diffusion_amd.model:create_latent_diffusion",
    "torch.optim.AdamW": "micro_diffusion_amd.trainer:FusedAdamW",
    "composer.optim.CosineAnnealingWithWarmupScheduler": "micro_diffusion_amd.trainer:LRSchedule",
    "composer.optim.ConstantScheduler": "micro_diffusion_amd.trainer:LRSchedule",
    "composer.optim.ConstantWithWarmupScheduler": "micro_diffusion_amd.trainer:LRSchedule",
    "micro_diffusion.datasets.latents_loader.build_streaming_latents_dataloader":
        "micro_diffusion_amd.data:build_streaming_latents_dataloader",
    "composer.Trainer": "micro_diffusion_amd.trainer:Trainer",
    "composer.loggers.TensorboardLogger": None,
    "composer.loggers.wandb_logger.WandBLogger": None,
    "composer.callbacks.speed_monitor.SpeedMonitor": None,
    "composer.callbacks.lr_monitor.LRMonitor": None,
    "composer.callbacks.runtime_estimator.RuntimeEstimator": None,
    "composer.callbacks.OptimizerMonitor": None,
    "micro_diffusion.models.callbacks.LogDiffusionImages": None,
    "micro_diffusion.models.callbacks.NaNCatcher": None,
    "diffusion.algorithms.ema.EMA": None,
}


def _parse_scalar(text: str) -> Any:
    return yaml.safe_load(text)


def apply_overrides(cfg: Dict[str, Any], overrides: List[str]) -> None:
    """Hydra-style dot-list overrides: `a.b.c=value` (value parsed as YAML), `+a.b=value` adds a new key."""
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                node[p] = {}
            node = node[p]
        node[parts[-1]] = _parse_scalar(val)


_INTERP = re.compile(r"\$\{([^}]+)\}")
_FLOATISH = re.compile(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+")


def coerce_numbers(cfg: Any) -> Any:
    """PyYAML (YAML 1.1) reads `8e-5` as a string; hydra/OmegaConf reads a float.  Match the latter."""
    if isinstance(cfg, dict):
        return {k: coerce_numbers(v) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [coerce_numbers(v) for v in cfg]
    if isinstance(cfg, str) and _FLOATISH.fullmatch(cfg.strip()):
        return float(cfg)
    return cfg


def _lookup(root: Dict[str, Any], dotted: str) -> Any:
    node: Any = root
    for p in dotted.split("."):
        node = node[p]
    return node


def resolve(cfg: Any, root: Dict[str, Any] = None) -> Any:
    """Resolve `${path.to.key}` references (whole-value references keep their type)."""
    root = cfg if root is None else root
    if isinstance(cfg, dict):
        return {k: resolve(v, root) for k, v in cfg.items()}
    if isinstance(cfg, list):
        return [resolve(v, root) for v in cfg]
    if isinstance(cfg, str):
        m = _INTERP.fullmatch(cfg.strip())
        if m:
            return resolve(_lookup(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(resolve(_lookup(root, mm.group(1)), root)), cfg)
    return cfg


def load_config(config_path: str, config_name: str, overrides: List[str] = ()) -> Dict[str, Any]:
    fn = os.path.join(config_path, config_name)
    if not os.path.exists(fn) and not fn.endswith((".yaml", ".yml")):
        fn += ".yaml"
    with open(fn) as fh:
        cfg = coerce_numbers(yaml.safe_load(fh))
    apply_overrides(cfg, list(overrides))
    return coerce_numbers(resolve(cfg))


def locate(target: str):
    native = TARGETS.get(target, target)
    if native is None:
        return None
    if ":" in native:
        mod, attr = native.split(":")
    else:
        mod, attr = native.rsplit(".", 1)
    return getattr(importlib.import_module(mod), attr)


def instantiate(node: Dict[str, Any], **extra):
    """hydra.utils.instantiate for the subset the reference uses (flat kwargs, `_target_`)."""
    node = dict(node)
    target = node.pop("_target_")
    fn = locate(target)
    if fn is None:
        return None
    node.update(extra)
    return fn(**node)
