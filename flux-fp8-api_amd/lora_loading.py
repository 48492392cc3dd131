"""LoRA fuse / unfuse into the (fp8) flow weights -- the device-side restatement of the reference's
lora_loading.py:476-753 (aredden/flux-fp8-api), config (5) of the benchmark plan.

In scope: dequantise fp8 -> fp32, dW = scale * (alpha/rank) * B @ A in fp32
(including the reference's "uneven rank" chunk-sum for fused qkv), add, round to the flow dtype,
re-quantise with a fresh amax/scale (F8Linear.set_weight_tensor semantics) -- all on the GPU via
libfluxmi (fluxmi_lora_fuse_f8).  Input scales are NOT recalibrated, exactly like the reference.

Accepted inputs: a dict with BFL-dotted keys `<module>.lora_A.weight / .lora_B.weight / .alpha`
(lora_loading.py:608-612 skips conversion for dicts), a LoraWeights, or a safetensors path holding such
keys or kohya `lora_unet_*` keys (converted below).  The diffusers `transformer.*` key conversion
(lora_loading.py:35-463) is host-side dict renaming outside the hot path: not built.
"""
from __future__ import annotations

import re
from typing import Optional, OrderedDict, Tuple, TypeAlias, Union

import torch
from torch import nn

from float8_quantize import F8Linear
from fluxmi import ops

path_regex = re.compile(r"/|\\")
StateDict: TypeAlias = OrderedDict[str, torch.Tensor]


class LoraWeights:  # reference lora_loading.py:21-32
    def __init__(self, weights: StateDict, path: str, name: str = None, scale: float = 1.0) -> None:
        self.path = path
        self.weights = weights
        self.name = name if name else path_regex.split(path)[-1]
        self.scale = scale


def get_module_for_key(key: str, model) -> nn.Module:
    module = model
    for part in key.split("."):
        module = getattr(module, part)
    return module


def get_lora_for_key(key: str, lora_weights: dict) -> Optional[Tuple[torch.Tensor, torch.Tensor, Optional[float]]]:
    prefix = key.split(".lora")[0]
    lora_A = lora_weights.get(f"{prefix}.lora_A.weight")
    lora_B = lora_weights.get(f"{prefix}.lora_B.weight")
    alpha = lora_weights.get(f"{prefix}.alpha")
    if lora_A is None or lora_B is None:
        return None
    return lora_A, lora_B, alpha


def _kohya_to_bfl(sd: dict) -> dict:
    """`lora_unet_double_blocks_0_img_attn_qkv.lora_down.weight` -> `double_blocks.0.img_attn.qkv.lora_A.weight`."""
    out = {}
    for k, v in sd.items():
        if not k.startswith("lora_unet_"):
            out[k] = v
            continue
        stem, _, leaf = k[len("lora_unet_"):].partition(".")
        stem = re.sub(r"^(double_blocks|single_blocks)_(\d+)_", r"\1.\2.", stem)
        for a, b in (("img_attn_", "img_attn."), ("txt_attn_", "txt_attn."), ("img_mlp_", "img_mlp."), ("txt_mlp_", "txt_mlp."),
                     ("img_mod_", "img_mod."), ("txt_mod_", "txt_mod."), ("modulation_", "modulation.")):
            stem = stem.replace(a, b)
        leaf = leaf.replace("lora_down", "lora_A").replace("lora_up", "lora_B")
        out[f"{stem}.{leaf}"] = v
    return out


def _keys_without_ab(lora_weights: dict):
    return sorted({k.replace(".lora_A.weight", "").replace(".lora_B.weight", "").replace(".lora_A", "").replace(".lora_B", "")
                   .replace(".alpha", "") for k in lora_weights.keys()})


def _resolve(lora_path):
    if isinstance(lora_path, LoraWeights):
        return lora_path.weights, lora_path.scale
    if isinstance(lora_path, dict):
        return lora_path, None
    from safetensors.torch import load_file

    sd = load_file(lora_path, "cpu")
    if any(k.startswith("transformer.") for k in sd):
        raise NotImplementedError("diffusers-format LoRA key conversion is outside the hot path (SURVEY.md §2.1)")
    return _kohya_to_bfl(sd), None


def _prescaled_A(lora_A, lora_B, alpha, device):
    """w_up = lora_A (fp32) * alpha / rank  (reference lora_loading.py:519-530); returns (A', n_chunks, rank)."""
    rank = lora_B.shape[1]
    if alpha is None:
        alpha = rank
    a = lora_A.to(dtype=torch.float32, device=device)
    if alpha != rank:
        a = a * alpha / rank
    uneven = lora_B.shape[1] != lora_A.shape[0]
    chunks = int(lora_A.shape[0] / lora_B.shape[1]) if uneven else 1
    return a.contiguous(), chunks, rank


@torch.inference_mode()
def _fuse_into(module: nn.Module, lora_sd, lora_scale: float):
    lora_A, lora_B, alpha = lora_sd
    if isinstance(module, F8Linear):
        dev = module.float8_data.device
        a, chunks, _ = _prescaled_A(lora_A, lora_B, alpha, dev)
        b = lora_B.to(dtype=torch.float32, device=dev).contiguous()
        ops.lora_fuse_f8(module.float8_data, module.scale, module.scale_reciprocal, b, a, lora_scale, n_chunks=chunks)
    else:  # un-quantised nn.Linear: one-off weight surgery in fp32 on the device (plumbing, not hot path)
        w = module.weight.data
        a, chunks, _ = _prescaled_A(lora_A, lora_B, alpha, w.device)
        b = lora_B.to(dtype=torch.float32, device=w.device)
        delta = sum(lora_scale * torch.mm(b, c) for c in a.chunk(chunks, dim=0))
        module.weight.data = (w.float() + delta).to(w.dtype)


@torch.inference_mode()
def apply_lora_to_model(model, lora_path, lora_scale: float = 1.0, return_lora_resolved: bool = False):
    """reference lora_loading.py:634-693."""
    lora_weights, _ = _resolve(lora_path)
    for key in _keys_without_ab(lora_weights):
        lora_sd = get_lora_for_key(key, lora_weights)
        if lora_sd is None:
            continue
        _fuse_into(get_module_for_key(key, model), lora_sd, lora_scale)
    if hasattr(model, "rebind_weights"):
        model.rebind_weights()
    if return_lora_resolved:
        return model, lora_weights
    return model


@torch.inference_mode()
def remove_lora_from_module(model, lora_path, lora_scale: float = 1.0):
    """reference lora_loading.py:696-753: subtract the same delta and re-quantise (lossy through fp8, as there)."""
    lora_weights, stored_scale = _resolve(lora_path)
    if stored_scale is not None:
        lora_scale = stored_scale
    for key in _keys_without_ab(lora_weights):
        lora_sd = get_lora_for_key(key, lora_weights)
        if lora_sd is None:
            continue
        _fuse_into(get_module_for_key(key, model), lora_sd, -lora_scale)
    if hasattr(model, "rebind_weights"):
        model.rebind_weights()
    return model
