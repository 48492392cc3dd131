"""CPU restatement of the reference VAE decoder and encoder (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Functional, state-dict driven torch code following modules/autoencoder.py of aredden/flux-fp8-api:
  swish :19-20 | AttnBlock :23-52 | ResnetBlock :55-93 | Downsample :95-107 | Upsample :110-120 | Encoder :123-200 | Decoder :203-283 |
  DiagonalGaussian :286-299 | AutoEncoder.encode / decode :326-332
`autocast=True` reproduces what flux_pipeline.py:431-434 runs (torch.autocast(bf16) around ae.decode) with explicit casts:
convolutions and SDPA see bf16 inputs/weights and return bf16; GroupNorm and the swish after it run in fp32.
Pinned by oracle/gen_golden_vae.py against the unmodified reference module (fp32: bit-equal; autocast: equal to the reference under
torch.autocast("cpu", bfloat16) up to the CPU autocast policy) -> tests/golden/g8_vae.safetensors.
"""
import torch
import torch.nn.functional as F


def _conv(sd, name, x, autocast, padding, stride=1):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if autocast:
        return F.conv2d(x.to(torch.bfloat16), w.to(torch.bfloat16), b.to(torch.bfloat16), padding=padding, stride=stride)
    return F.conv2d(x.float(), w.float(), b.float(), padding=padding, stride=stride)


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


def encode_moments(sd, params, x, autocast=True):
    """Encoder.forward (:176-200): [B, 3, H, W] -> [B, 2*z_channels, H/8, W/8] (mean | logvar)."""
    nres = len(params["ch_mult"])
    h = _conv(sd, "encoder.conv_in", x, autocast, 1)
    for lvl in range(nres):
        for ib in range(params["num_res_blocks"]):
            h = resnet_block(sd, f"encoder.down.{lvl}.block.{ib}", h, autocast)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)  # :104-105 (asymmetric: right / bottom only)
            h = _conv(sd, f"encoder.down.{lvl}.downsample.conv", h, autocast, 0, stride=2)
    h = resnet_block(sd, "encoder.mid.block_1", h, autocast)
    h = attn_block(sd, "encoder.mid.attn_1", h, autocast)
    h = resnet_block(sd, "encoder.mid.block_2", h, autocast)
    h = _gn_swish(sd, "encoder.norm_out", h)
    return _conv(sd, "encoder.conv_out", h, autocast, 1)


def encode(sd, params, x, noise=None, autocast=True):
    """AutoEncoder.encode (:326-329) with DiagonalGaussian (:292-299); `noise` stands in for torch.randn_like(mean) (None: mean only)."""
    mean, logvar = torch.chunk(encode_moments(sd, params, x, autocast), 2, dim=1)
    z = mean if noise is None else mean + torch.exp(0.5 * logvar.float()) * noise.to(mean.dtype)
    return params["scale_factor"] * (z - params["shift_factor"])


# ---- real-size fixture (oracle/gen_golden_vae_full.py, tests/test_engine_gpu.py::test_vae_full_size_matches_reference_fixture) ----------
FULL_PARAMS = dict(resolution=256, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, z_channels=16,
                   scale_factor=0.3611, shift_factor=0.1159)  # reference util.py:99-110


def synth_state_dict(shapes, seed=7):
    """Deterministic weights for a VAE with the given {key: shape} (BFL `ae.sft` layout), bf16-representable (ae_dtype = bfloat16):
    convolutions ~ N(0, 1/fan_in), norm weights 1 + 0.1 N(0,1), biases 0.05 N(0,1).  Filled in sorted key order from ONE seeded CPU
    generator, so the reference module and the native module get identical tensors."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if "norm" in k and k.endswith(".weight"):
            t = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g) / max(fan_in, 1) ** 0.5
        sd[k] = t.to(torch.bfloat16).float()
    return sd


def full_inputs(seed=11):
    """(latent z [1,16,32,32] fp32, image x [1,3,256,256] in [-1,1], bf16-representable as the pipeline hands it over)"""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(1, 16, 32, 32, generator=g)
    x = (torch.rand(1, 3, 256, 256, generator=g) * 2 - 1).to(torch.bfloat16).float()
    return z, x
