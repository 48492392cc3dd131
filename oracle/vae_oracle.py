"""CPU restatement of the reference VAE decoder (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Functional, state-dict driven torch code following modules/autoencoder.py of aredden/flux-fp8-api:
  swish :19-20 | AttnBlock :23-52 | ResnetBlock :55-93 | Upsample :110-120 | Decoder :203-283 | AutoEncoder.decode :330-332
`autocast=True` reproduces what flux_pipeline.py:431-434 runs (torch.autocast(bf16) around ae.decode) with explicit casts:
convolutions and SDPA see bf16 inputs/weights and return bf16; GroupNorm and the swish after it run in fp32.
Pinned by oracle/gen_golden.py against the unmodified reference module (fp32: bit-equal; autocast: equal to the reference under
torch.autocast("cpu", bfloat16) up to the CPU autocast policy) -> tests/golden/g8_vae.safetensors.
"""
import torch
import torch.nn.functional as F


def _conv(sd, name, x, autocast, padding):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if autocast:
        return F.conv2d(x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16), padding=padding)
    return F.conv2d(x.float(), w.float(), b.float(), padding=padding)


def _gn_swish(sd, name, x, swish=True):
    h = F.group_norm(x.float(), 32, sd[name + ".weight"].float(), sd[name + ".bias"].float(), eps=1e-6)  # :28-30,62-70
    return h * torch.sigmoid(h) if swish else h  # :19-20


def resnet_block(sd, pre, x, autocast):  # :79-92
    h = _conv(sd, pre + ".conv1", _gn_swish(sd, pre + ".norm1", x), autocast, 1)
    h = _conv(sd, pre + ".conv2", _gn_swish(sd, pre + ".norm2", h), autocast, 1)
    if (pre + ".nin_shortcut.weight") in sd:
        x = _conv(sd, pre + ".nin_shortcut", x, autocast, 0)
    return x + h


def attn_block(sd, pre, x, autocast):  # :37-52
    h = _gn_swish(sd, pre + ".norm", x, swish=False)
    q, k, v = (_conv(sd, pre + "." + n, h, autocast, 0) for n in "qkv")
    b, c, hh, ww = q.shape
    q, k, v = (t.reshape(b, 1, c, hh * ww).transpose(2, 3).contiguous() for t in (q, k, v))  # b 1 (h w) c
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(2, 3).reshape(b, c, hh, ww)
    return x + _conv(sd, pre + ".proj_out", o, autocast, 0)


def decode(sd, params, z, autocast=True):
    """AutoEncoder.decode (:330-332) + Decoder.forward (:261-283).  params: dict with ch_mult, num_res_blocks, scale/shift_factor."""
    z = z.float() / params["scale_factor"] + params["shift_factor"]
    nres = len(params["ch_mult"])
    h = _conv(sd, "decoder.conv_in", z, autocast, 1)
    h = resnet_block(sd, "decoder.mid.block_1", h, autocast)
    h = attn_block(sd, "decoder.mid.attn_1", h, autocast)
    h = resnet_block(sd, "decoder.mid.block_2", h, autocast)
    for lvl in reversed(range(nres)):
        for ib in range(params["num_res_blocks"] + 1):
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{ib}", h, autocast)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # :117-119
            h = _conv(sd, f"decoder.up.{lvl}.upsample.conv", h, autocast, 1)
    h = _gn_swish(sd, "decoder.norm_out", h)
    return _conv(sd, "decoder.conv_out", h, autocast, 1)
