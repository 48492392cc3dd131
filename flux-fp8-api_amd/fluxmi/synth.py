"""Synthetic Flux checkpoints and inputs (no real weights exist offline).

Produces a BFL-format state dict (key names / shapes of SURVEY.md Appendix D, i.e. what
`util.load_flow_model` feeds to `Flux.load_state_dict`) from a seed, plus the seeded synthetic
request tensors of SURVEY.md §8(d).  Pure torch, CPU by default so that the bytes are identical on
every machine; pass device="cuda" for the full 12 B-parameter bench model (device RNG, not portable,
parity tests never use it).
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def _linear(sd, name, n_out, n_in, gen, device, dtype, gain=1.0, bias=True, outliers=True):
    bound = gain * math.sqrt(3.0 / n_in)
    w = (torch.rand(n_out, n_in, generator=gen, device=device, dtype=torch.float32) * 2 - 1) * bound
    if outliers and n_out * n_in >= 4096:
        # 0.1 % of the entries x20: exercises the amax->scale clamp and fp8 saturation branches
        n_spike = max(1, (n_out * n_in) // 1000)
        idx = torch.randint(0, n_out * n_in, (n_spike,), generator=gen, device=device)
        w.view(-1)[idx] *= 20.0
    sd[name + ".weight"] = w.to(dtype)
    if bias:
        b = (torch.rand(n_out, generator=gen, device=device, dtype=torch.float32) * 2 - 1) / math.sqrt(n_in)
        sd[name + ".bias"] = b.to(dtype)


def make_state_dict(params, seed: int = 0, dtype=torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    """params: anything with the FluxParams fields (modules.flux_model.FluxParams)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    H = params.hidden_size
    mlp = int(H * params.mlp_ratio)
    hd = H // params.num_heads
    sd: Dict[str, torch.Tensor] = {}
    L = lambda *a, **k: _linear(sd, *a, gen=gen, device=device, dtype=dtype, **k)
    L("img_in", H, params.in_channels)
    for emb, d_in in (("time_in", 256), ("vector_in", params.vec_in_dim)) + (
        (("guidance_in", 256),) if params.guidance_embed else ()
    ):
        L(emb + ".in_layer", H, d_in)
        L(emb + ".out_layer", H, H)
    L("txt_in", H, params.context_in_dim)

    def qk_scales(prefix):
        for nm in ("query_norm", "key_norm"):
            s = 1.0 + 0.1 * torch.randn(hd, generator=gen, device=device, dtype=torch.float32)
            sd[f"{prefix}.{nm}.scale"] = s.to(dtype)

    for i in range(params.depth):
        p = f"double_blocks.{i}"
        for s in ("img", "txt"):
            L(f"{p}.{s}_mod.lin", 6 * H, H, gain=0.5)
            L(f"{p}.{s}_attn.qkv", 3 * H, H, bias=params.qkv_bias)
            qk_scales(f"{p}.{s}_attn.norm")
            L(f"{p}.{s}_attn.proj", H, H)
            L(f"{p}.{s}_mlp.0", mlp, H)
            L(f"{p}.{s}_mlp.2", H, mlp)
    for i in range(params.depth_single_blocks):
        p = f"single_blocks.{i}"
        L(f"{p}.linear1", 3 * H + mlp, H)
        L(f"{p}.linear2", H, H + mlp)
        qk_scales(f"{p}.norm")
        L(f"{p}.modulation.lin", 3 * H, H, gain=0.5)
    L("final_layer.linear", params.in_channels, H)
    L("final_layer.adaLN_modulation.1", 2 * H, H, gain=0.5)
    return sd


def make_inputs(params, height: int, width: int, txt_len: int, batch: int = 1, seed: int = 0,
                dtype=torch.bfloat16, real_tokens: int = 32):
    """Seeded request tensors: packed N(0,1) latent noise, T5-like `txt` whose rows
    >= real_tokens repeat one "pad" row, CLIP-like pooled `y`, position ids per
    flux_pipeline.py:280-292 / flux_emphasis.py:433-439.  All on CPU in the flow dtype."""
    gen = torch.Generator().manual_seed(1000 + seed)
    h2, w2 = math.ceil(height / 16), math.ceil(width / 16)
    img = torch.randn(batch, h2 * w2, params.in_channels, generator=gen).to(dtype)
    txt = 0.1 * torch.randn(batch, txt_len, params.context_in_dim, generator=gen)
    if txt_len > real_tokens:
        txt[:, real_tokens:] = txt[:, real_tokens:real_tokens + 1]
    txt = txt.to(dtype)
    y = torch.randn(batch, params.vec_in_dim, generator=gen).to(dtype)
    img_ids = torch.zeros(h2, w2, 3, dtype=dtype)
    img_ids[..., 1] += torch.arange(h2, dtype=dtype)[:, None]
    img_ids[..., 2] += torch.arange(w2, dtype=dtype)[None, :]
    img_ids = img_ids[None].repeat(batch, 1, 1, 1).flatten(1, 2)
    txt_ids = torch.zeros(batch, txt_len, 3, dtype=dtype)
    return dict(img=img, img_ids=img_ids, txt=txt, txt_ids=txt_ids, y=y)
