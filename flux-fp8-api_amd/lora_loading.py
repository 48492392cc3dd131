"""LoRA fuse / unfuse into the (fp8) flow weights -- the device-side restatement of the reference's
lora_loading.py:476-753 (aredden/flux-fp8-api), config (5) of the benchmark plan.

In scope: dequantise fp8 -> fp32, dW = scale * (alpha/rank) * B @ A in fp32
(including the reference's "uneven rank" chunk-sum for fused qkv), add, round to the flow dtype,
re-quantise with a fresh amax/scale (F8Linear.set_weight_tensor semantics) -- all on the GPU via
libfluxmi (fluxmi_lora_fuse_f8).  Input scales are NOT recalibrated, exactly like the reference.

Accepted inputs: a dict with BFL-dotted keys `<module>.lora_A.weight / .lora_B.weight / .alpha`
(lora_loading.py:608-612 skips conversion for dicts), a LoraWeights, or a safetensors path holding such
keys, kohya `lora_unet_*` keys, or diffusers `transformer.*` keys (both converted below; the diffusers
conversion is pinned to the reference's convert_diffusers_to_flux_transformer_checkpoint by
tests/golden/g12_lora_diffusers.safetensors, quirks included).
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


# ---- diffusers ("transformer.*") LoRA files -> BFL module names ------------------------------------------------------------------------
# Behaviour of the reference's convert_diffusers_to_flux_transformer_checkpoint (lora_loading.py:62-432), restated as tables:
#  * a renamed layer takes EVERY key that contains the diffusers stem as a substring (so `.alpha` and bias keys travel with it);
#  * to_q / to_k / to_v (and add_*_proj) of a double block are concatenated along dim 0 into the fused qkv layer's "uneven rank" form
#    A [3r, K], B [3N', r] (which apply_lora then chunk-sums, lora_loading.py:533-541); a missing member is zero-filled with the shape of
#    the first member found; their `.alpha` keys are NOT consumed (the reference leaves them among the "unexpected keys");
#  * a single block needs all of to_q / to_k / to_v / proj_mlp (the reference pops them unconditionally: KeyError otherwise) and becomes
#    linear1 with A [4r, K], B [3H + mlp, r];
#  * norm_out.linear -> final_layer.adaLN_modulation.1 WITHOUT swapping (scale, shift) (the reference defines swap_scale_shift and
#    never calls it).
_DIFFUSERS_TOP = [
    ("time_text_embed.timestep_embedder.linear_1", "time_in.in_layer", False), ("time_text_embed.text_embedder.linear_1", "vector_in.in_layer", False),
    ("time_text_embed.text_embedder.linear_2", "vector_in.out_layer", False),
    ("time_text_embed.guidance_embedder.linear_1", "guidance_in.in_layer", True), ("time_text_embed.guidance_embedder.linear_2", "guidance_in.out_layer", True),
    ("context_embedder", "txt_in", False), ("x_embedder", "img_in", False),
]
_DIFFUSERS_DOUBLE_PRE = [("norm1.linear", "img_mod.lin"), ("norm1_context.linear", "txt_mod.lin")]
_DIFFUSERS_DOUBLE_POST = [
    ("attn.norm_q", "img_attn.norm.query_norm.scale"), ("attn.norm_k", "img_attn.norm.key_norm.scale"),
    ("attn.norm_added_q", "txt_attn.norm.query_norm.scale"), ("attn.norm_added_k", "txt_attn.norm.key_norm.scale"),
    ("ff.net.0.proj", "img_mlp.0"), ("ff.net.2", "img_mlp.2"), ("ff_context.net.0.proj", "txt_mlp.0"), ("ff_context.net.2", "txt_mlp.2"),
    ("attn.to_out.0", "img_attn.proj"), ("attn.to_add_out", "txt_attn.proj"),
]


def _move_layer(out: dict, sd: dict, stem: str, new_stem: str) -> None:
    hit = [k for k in sd if stem in k]
    for k in hit:
        out[k.replace(stem, new_stem)] = sd.pop(k)


def convert_diffusers_to_flux_transformer_checkpoint(diffusers_state_dict, num_layers, num_single_layers, has_guidance=True, prefix=""):
    """diffusers-format Flux LoRA keys -> `<BFL module>.lora_A.weight / .lora_B.weight` (same name, arguments and result as the
    reference's function; consumes `diffusers_state_dict` like it does)."""
    sd, out = diffusers_state_dict, {}
    for stem, new, guidance_only in _DIFFUSERS_TOP:
        if not guidance_only or has_guidance:
            _move_layer(out, sd, prefix + stem, new)
    dtype = device = None  # (carried from the double blocks into the single blocks' zero fills, as in the reference)
    for i in range(num_layers):
        bp = f"{prefix}transformer_blocks.{i}."
        for stem, new in _DIFFUSERS_DOUBLE_PRE:
            _move_layer(out, sd, bp + stem, f"double_blocks.{i}.{new}")
        found, shape = {}, {"img": None, "txt": None}
        for comp in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"):
            ka, kb = f"{bp}attn.{comp}.lora_A.weight", f"{bp}attn.{comp}.lora_B.weight"
            if ka in sd and kb in sd:
                a, b = sd.pop(ka), sd.pop(kb)
                found[comp] = (a, b)
                stream = "txt" if comp.startswith("add_") else "img"
                if shape[stream] is None:
                    shape[stream] = (a.shape, b.shape)
                    dtype, device = a.dtype, a.device
        for stream, comps in (("img", ("to_q", "to_k", "to_v")), ("txt", ("add_q_proj", "add_k_proj", "add_v_proj"))):
            if shape[stream] is None:
                continue
            parts = [found.get(c) or (torch.zeros(shape[stream][0], dtype=dtype, device=device), torch.zeros(shape[stream][1], dtype=dtype, device=device))
                     for c in comps]
            out[f"double_blocks.{i}.{stream}_attn.qkv.lora_A.weight"] = torch.cat([pa for pa, _ in parts], dim=0)
            out[f"double_blocks.{i}.{stream}_attn.qkv.lora_B.weight"] = torch.cat([pb for _, pb in parts], dim=0)
        for stem, new in _DIFFUSERS_DOUBLE_POST:
            _move_layer(out, sd, bp + stem, f"double_blocks.{i}.{new}")
    for i in range(num_single_layers):
        bp = f"{prefix}single_transformer_blocks.{i}."
        _move_layer(out, sd, bp + "norm.linear", f"single_blocks.{i}.modulation.lin")
        parts = [(sd.pop(f"{bp}{c}.lora_A.weight"), sd.pop(f"{bp}{c}.lora_B.weight")) for c in ("attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp")]
        out[f"single_blocks.{i}.linear1.lora_A.weight"] = torch.cat([pa for pa, _ in parts], dim=0)
        out[f"single_blocks.{i}.linear1.lora_B.weight"] = torch.cat([pb for _, pb in parts], dim=0)
        _move_layer(out, sd, bp + "proj_out", f"single_blocks.{i}.linear2")
    _move_layer(out, sd, prefix + "proj_out", "final_layer.linear")  # (weight and bias stems coincide)
    _move_layer(out, sd, prefix + "norm_out.linear", "final_layer.adaLN_modulation.1")
    return out


def resolve_lora_state_dict(lora_weights: dict, has_guidance: bool = True):
    """reference lora_loading.py:580-606: file contents -> (module stems, BFL-keyed weights); 19 / 38 blocks are hard-coded there too."""
    if any(k.startswith("transformer.") for k in lora_weights):
        lora_weights = convert_diffusers_to_flux_transformer_checkpoint(dict(lora_weights), 19, 38, has_guidance=has_guidance, prefix="transformer.")
    else:
        lora_weights = _kohya_to_bfl({k: v for k, v in lora_weights.items() if "lora" in k})
    return _keys_without_ab(lora_weights), lora_weights


def _keys_without_ab(lora_weights: dict):
    return sorted({k.replace(".lora_A.weight", "").replace(".lora_B.weight", "").replace(".lora_A", "").replace(".lora_B", "")
                   .replace(".alpha", "") for k in lora_weights.keys()})


def _resolve(lora_path, has_guidance: bool = True):
    if isinstance(lora_path, LoraWeights):
        return lora_path.weights, lora_path.scale
    if isinstance(lora_path, dict):
        return lora_path, None
    from safetensors.torch import load_file

    return resolve_lora_state_dict(load_file(lora_path, "cpu"), has_guidance)[1], None


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
    lora_weights, _ = _resolve(lora_path, bool(getattr(getattr(model, "params", None), "guidance_embed", True)))
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
    lora_weights, stored_scale = _resolve(lora_path, bool(getattr(getattr(model, "params", None), "guidance_embed", True)))
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
