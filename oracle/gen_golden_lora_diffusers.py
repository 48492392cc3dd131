#!/usr/bin/env python
"""Fixture for the diffusers -> BFL LoRA key conversion: runs the UNMODIFIED reference's convert_diffusers_to_flux_transformer_checkpoint
(lora_loading.py:62-432, through resolve_lora_state_dict :580-606) on synthetic diffusers-format LoRA dicts and stores input + output.
Build container only (needs /root/reference):  python oracle/gen_golden_lora_diffusers.py  ->  tests/golden/g12_lora_diffusers.safetensors
Two inputs: "full" = every key family the converter knows (embedders, both modulations, q/k/v + add_q/k/v, norms, both MLPs, both output
projections, single blocks, final layer, .alpha and bias keys) on a flux-dev-shaped tree (19 + 38 blocks: the reference hard-codes the
depth and POPS the single-block q/k/v/mlp keys unconditionally); "sparse" = a realistic attention-only LoRA where some double blocks lack
to_k or all add_*_proj (zero-filled members / skipped streams).  Tensors are tiny (rank 2, width 6-10): only names and row order matter."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import torch
from safetensors.torch import save_file

import ref_shims

f8q, fm, rutil = ref_shims.import_reference()
import lora_loading as ref_ll  # the reference's (ref_shims put /root/reference first on sys.path)

assert ref_ll.__file__.startswith(ref_shims.REFERENCE_ROOT), ref_ll.__file__


def synth(sparse: bool, seed: int):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    r = 2

    def lora(stem, n_out=8, n_in=6, alpha=True, bias=False):
        sd[f"transformer.{stem}.lora_A.weight"] = torch.randn(r, n_in, generator=g)
        sd[f"transformer.{stem}.lora_B.weight"] = torch.randn(n_out, r, generator=g)
        if alpha:
            sd[f"transformer.{stem}.alpha"] = torch.tensor(float(1 + len(sd) % 5))
        if bias:
            sd[f"transformer.{stem}.lora_B.bias"] = torch.randn(n_out, generator=g)

    if not sparse:
        for stem in ("time_text_embed.timestep_embedder.linear_1", "time_text_embed.text_embedder.linear_1", "time_text_embed.text_embedder.linear_2",
                     "time_text_embed.guidance_embedder.linear_1", "time_text_embed.guidance_embedder.linear_2", "context_embedder", "x_embedder"):
            lora(stem)
    for i in range(19):
        b = f"transformer_blocks.{i}"
        comps = ["to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"]
        if sparse:
            if i % 3 == 1:
                comps.remove("to_k")
            if i % 4 == 2:
                comps = [c for c in comps if not c.startswith("add_")]
            if i % 5 == 4:
                comps = []
        for c in comps:
            lora(f"{b}.attn.{c}", alpha=not sparse)
        if not sparse:
            for stem in ("norm1.linear", "norm1_context.linear", "ff.net.0.proj", "ff.net.2", "ff_context.net.0.proj", "ff_context.net.2",
                         "attn.to_out.0", "attn.to_add_out"):
                lora(f"{b}.{stem}", bias=(stem == "attn.to_out.0" and i == 0))
        elif i % 2 == 0:
            lora(f"{b}.attn.to_out.0", alpha=False)
    for i in range(38):
        b = f"single_transformer_blocks.{i}"
        for c in ("attn.to_q", "attn.to_k", "attn.to_v"):
            lora(f"{b}.{c}", alpha=False)
        lora(f"{b}.proj_mlp", n_out=10, alpha=False)
        if not sparse:
            lora(f"{b}.norm.linear")
            lora(f"{b}.proj_out")
    if not sparse:
        lora("proj_out", bias=True)
        lora("norm_out.linear")
    return sd


out = {}
for name, sparse, seed in (("full", False, 1), ("sparse", True, 2)):
    inp = synth(sparse, seed)
    for k, v in inp.items():
        out[f"{name}:in:{k}"] = v.clone()
    keys, conv = ref_ll.resolve_lora_state_dict({k: v.clone() for k, v in inp.items()}, has_guidance=True)
    for k, v in conv.items():
        out[f"{name}:out:{k}"] = v.clone()
    out[f"{name}:n_keys_without_ab"] = torch.tensor(len(keys))
    print(f"{name}: {len(inp)} diffusers keys -> {len(conv)} BFL keys, {len(keys)} module stems")
# and without the guidance embedder (flux-schnell): the guidance_embedder keys stay unconverted
inp = synth(False, 1)
keys, conv = ref_ll.resolve_lora_state_dict({k: v.clone() for k, v in inp.items()}, has_guidance=False)
for k, v in conv.items():
    out[f"noguidance:out:{k}"] = v.clone()
path = os.path.join(ROOT, "tests", "golden", "g12_lora_diffusers.safetensors")
save_file({k: v.contiguous() for k, v in out.items()}, path)
print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "tensors")
