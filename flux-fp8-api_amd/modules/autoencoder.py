"""VAE of the reference config schema (modules/autoencoder.py of aredden/flux-fp8-api) on MI355X.

SURVEY.md §8(f) row 1: `decode` is the step right after the denoise loop, `encode` the img2img entry before it.  `AutoEncoder(params)`
keeps the reference's module tree and state-dict keys (`decoder.conv_in`, `decoder.mid.block_1.norm1`, `decoder.up.3.block.0.conv1`,
`decoder.up.1.upsample.conv`, `encoder.down.0.downsample.conv`, ... so a BFL `ae.sft` loads with `load_state_dict`); both directions
run natively:

  * activations NHWC bf16; 3x3 convolutions = `fluxmi_im2col3x3` (the 2x nearest upsample folded into the gather) + the bf16 MFMA
    GEMM with the weight reordered once to [Cout][dy][dx][Cin]; 1x1 convolutions are plain GEMMs; residual adds ride in the GEMM's
    gate*y+x epilogue (gate = 1);
  * GroupNorm(32) + swish in fp32, rounded to bf16 once (`fluxmi_groupnorm`) -- what torch.autocast(bf16) computes
    (flux_pipeline.py:431-434): convolutions / SDPA in bf16, GroupNorm and the swish behind it in fp32;
  * the single 512-wide attention head of mid.attn_1: S = Q K^T (GEMM) -> fp32 row softmax (`fluxmi_softmax_rows`) -> P V (GEMM against
    V^T, which the v-projection GEMM produces directly by swapping its operands; v's bias is added after P V, rows of P sum to 1).

  * Downsample (encoder): the stride-2 window with zero pad on the right/bottom only is one more gather mode of `fluxmi_im2col3x3`;
  * DiagonalGaussian: mean + exp(logvar/2) * noise on the [B, 2z, h, w] moments (a 128 KB tensor at 1024^2 -- torch elementwise).

There is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import torch
from pydantic import BaseModel
from torch import Tensor, nn


class AutoEncoderParams(BaseModel):
    resolution: int
    in_channels: int
    ch: int
    out_ch: int
    ch_mult: list[int]
    num_res_blocks: int
    z_channels: int
    scale_factor: float
    shift_factor: float


def _gn(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class AttnBlock(nn.Module):  # reference :23-52 (parameters only; the math is AutoEncoder._attn)
    def __init__(self, c: int):
        super().__init__()
        self.norm = _gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, kernel_size=1) for _ in range(4))


class ResnetBlock(nn.Module):  # reference :55-93
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1, self.conv1 = _gn(cin), nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1)
        self.norm2, self.conv2 = _gn(cout), nn.Conv2d(cout, cout, kernel_size=3, stride=1, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0)


class Upsample(nn.Module):  # reference :110-120
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1)


class Downsample(nn.Module):  # reference :95-107 (stride 2, zero pad right/bottom only: AutoEncoder._conv3(mode=-2))
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, kernel_size=3, stride=2, padding=0)


class Encoder(nn.Module):  # reference :123-174 (module tree / state-dict keys)
    def __init__(self, resolution, in_channels, ch, ch_mult, num_res_blocks, z_channels):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            down = nn.Module()
            down.block, down.attn = block, nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels, kernel_size=3, stride=1, padding=1)


class DiagonalGaussian(nn.Module):  # reference :286-299
    def __init__(self, sample: bool = True, chunk_dim: int = 1):
        super().__init__()
        self.sample, self.chunk_dim = sample, chunk_dim

    def forward(self, z: Tensor, noise: Tensor | None = None) -> Tensor:
        mean, logvar = torch.chunk(z, 2, dim=self.chunk_dim)
        if not self.sample:
            return mean
        std = torch.exp(0.5 * logvar.float())  # autocast runs exp in fp32; mean (bf16) + fp32 -> fp32
        return mean + std * (torch.randn_like(mean) if noise is None else noise.to(mean.dtype))


class Decoder(nn.Module):  # reference :203-259 (module tree / state-dict keys)
    def __init__(self, ch, out_ch, ch_mult, num_res_blocks, in_channels, resolution, z_channels):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            up = nn.Module()
            up.block, up.attn = block, nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)


class AutoEncoder(nn.Module):
    def __init__(self, params: AutoEncoderParams):
        super().__init__()
        self.params = params
        self.encoder = Encoder(resolution=params.resolution, in_channels=params.in_channels, ch=params.ch, ch_mult=params.ch_mult,
                               num_res_blocks=params.num_res_blocks, z_channels=params.z_channels)
        self.reg = DiagonalGaussian()
        self.decoder = Decoder(resolution=params.resolution, in_channels=params.in_channels, ch=params.ch, out_ch=params.out_ch,
                               ch_mult=params.ch_mult, num_res_blocks=params.num_res_blocks, z_channels=params.z_channels)
        self.scale_factor, self.shift_factor = params.scale_factor, params.shift_factor
        self.encoder_loaded = True  # util.load_autoencoder clears it for decoder-only checkpoints
        self._wcache = {}
        self.implicit_conv = True  # 3x3 convolutions as implicit GEMMs (fluxmi_conv3x3); False = im2col + GEMM (same bits, tests)

    # ---- weight preparation (once per module): conv weight -> GEMM weight [N, K] bf16, K ordered (dy, dx, cin) --------------------
    def _w(self, conv: nn.Conv2d):
        key = id(conv)
        ent = self._wcache.get(key)
        if ent is None or ent[0] is not conv.weight or ent[1].device != conv.weight.device:
            w = conv.weight.detach()
            cin = w.shape[1]
            pad = (-cin) % 8  # the patch gather moves 8 channels at a time
            if pad:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad))
            w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.bfloat16).contiguous()
            b = conv.bias.detach().to(torch.bfloat16).contiguous()
            ent = (conv.weight, w2, b, pad)
            self._wcache[key] = ent
        return ent[1], ent[2], ent[3]

    def _conv3(self, x: Tensor, conv: nn.Conv2d, upsample: int = 1, resid: Tensor | None = None) -> Tensor:
        from fluxmi import _lib, ops

        w2, b, pad = self._w(conv)
        if pad:
            x = torch.nn.functional.pad(x, (0, pad))
        B, Hi, Wi, _ = x.shape
        # round 6: implicit GEMM -- the tile kernel gathers each K-step from the NHWC input itself; the [pixels, 9 C] patch matrix (2.4 - 4.8 GB
        # per 1024^2 convolution) exists only for the two convolutions whose channel counts do not tile (conv_in: 16 channels, conv_out: 3 outputs)
        if self.implicit_conv and ops.conv3x3_ok(x.shape[-1], w2.shape[0]):
            gate = self._ones(w2.shape[0], x.device) if resid is not None else None
            return ops.conv3x3(x, w2, b, upsample, resid=resid, gate=gate)
        if self.implicit_conv and resid is None and ops.conv3x3_ok(x.shape[-1], 128) and w2.shape[0] < 128:
            # few output channels (decoder.conv_out: 128 -> 3 at full resolution): zero rows up to one 128-column tile, slice afterwards -- 268 MB of
            # discarded outputs per 1024^2 image instead of a 2.4 GB patch matrix written and read back
            key = ("pad128", id(conv))
            ent = self._wcache.get(key)
            if ent is None or ent[0] is not conv.weight or ent[1].device != w2.device:
                wp = torch.zeros(128, w2.shape[1], dtype=w2.dtype, device=w2.device)
                wp[: w2.shape[0]] = w2
                bp = torch.zeros(128, dtype=b.dtype, device=b.device)
                bp[: b.shape[0]] = b
                ent = (conv.weight, wp, bp)
                self._wcache[key] = ent
            return ops.conv3x3(x, ent[1], ent[2], upsample)[..., : w2.shape[0]].contiguous()
        col = ops.im2col3x3(x, upsample)  # upsample: 1 | 2 (nearest 2x first) | -2 (stride 2, right/bottom zero pad)
        Ho, Wo = (Hi // 2, Wi // 2) if upsample == -2 else (Hi * upsample, Wi * upsample)
        return self._gemm(col, w2, b, resid).view(B, Ho, Wo, -1)

    def _conv1(self, x: Tensor, conv: nn.Conv2d, resid: Tensor | None = None) -> Tensor:
        w2, b, _ = self._w(conv)
        return self._gemm(x.reshape(-1, x.shape[-1]), w2, b, resid).view(*x.shape[:-1], -1)

    def _gemm(self, a: Tensor, w2: Tensor, bias: Tensor | None, resid: Tensor | None) -> Tensor:
        from fluxmi import _lib, ops

        if resid is None:
            return ops.linear(a, w2, bias)
        r = resid.reshape(-1, resid.shape[-1])
        out = torch.empty_like(r)
        ones = self._ones(w2.shape[0], a.device)
        return ops.linear(a, w2, bias, epilogue=_lib.EPI_GATE_RESID, gate=ones, resid=r, out=out)  # out = resid + 1 * (a @ w2^T + bias)

    def _ones(self, n, device):
        key = ("ones", n, str(device))
        if key not in self._wcache:
            self._wcache[key] = torch.ones(n, dtype=torch.bfloat16, device=device)
        return self._wcache[key]

    def _norm(self, x: Tensor, gn: nn.GroupNorm, swish: bool) -> Tensor:
        from fluxmi import ops

        B, H, W, C = x.shape
        key = id(gn)
        ent = self._wcache.get(key)
        if ent is None or ent[0] is not gn.weight or ent[1].device != x.device:
            ent = (gn.weight, gn.weight.detach().to(torch.bfloat16).contiguous(), gn.bias.detach().to(torch.bfloat16).contiguous())
            self._wcache[key] = ent
        return ops.groupnorm(x.view(B, H * W, C), ent[1], ent[2], swish=swish, eps=gn.eps).view(B, H, W, C)

    def _resnet(self, x: Tensor, blk: ResnetBlock) -> Tensor:  # reference :79-92
        h = self._conv3(self._norm(x, blk.norm1, True), blk.conv1)
        if blk.in_channels != blk.out_channels:
            x = self._conv1(x, blk.nin_shortcut)
        return self._conv3(self._norm(h, blk.norm2, True), blk.conv2, resid=x)

    def _attn(self, x: Tensor, blk: AttnBlock) -> Tensor:  # reference :37-52
        from fluxmi import ops

        B, H, W, C = x.shape
        hn = self._norm(x, blk.norm, False).view(B, H * W, C)
        wv, bv, _ = self._w(blk.v)
        out = torch.empty_like(hn)
        for b in range(B):
            q = self._conv1(hn[b], blk.q)                       # [P, C]
            k = self._conv1(hn[b], blk.k)
            vt = ops.linear(wv, hn[b].contiguous(), None)       # [C, P] = Wv . Xn^T = V^T (bias added after P V)
            S = ops.linear(q, k, None)                          # [P, P] = Q K^T
            Pm = ops.softmax_rows(S, float(C) ** -0.5)
            out[b] = ops.linear(Pm, vt, bv)                     # [P, C] = P V + b_v
        return self._conv1(out.view(B, H, W, C), blk.proj_out, resid=x)

    @torch.inference_mode()
    def decode(self, z: Tensor) -> Tensor:
        """z [B, z_channels, h, w] -> image [B, out_ch, 8h, 8w] (bf16), reference :330-332 + :261-283 under autocast(bf16)."""
        if not z.is_cuda:
            raise RuntimeError("fluxmi: AutoEncoder.decode needs the GPU (libfluxmi has no CPU path)")
        d = self.decoder
        h = (z.float() / self.scale_factor + self.shift_factor).permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()  # NHWC
        h = self._conv3(h, d.conv_in)
        h = self._resnet(h, d.mid.block_1)
        h = self._attn(h, d.mid.attn_1)
        h = self._resnet(h, d.mid.block_2)
        for i_level in reversed(range(d.num_resolutions)):
            for i_block in range(d.num_res_blocks + 1):
                h = self._resnet(h, d.up[i_level].block[i_block])
            if i_level != 0:
                h = self._conv3(h, d.up[i_level].upsample.conv, upsample=2)
        h = self._conv3(self._norm(h, d.norm_out, True), d.conv_out)
        return h.permute(0, 3, 1, 2).contiguous()

    @torch.inference_mode()
    def encode_moments(self, x: Tensor) -> Tensor:
        """image [B, in_channels, H, W] in [-1, 1] -> moments [B, 2*z_channels, H/8, W/8] (bf16): Encoder.forward, reference :176-200."""
        if not x.is_cuda:
            raise RuntimeError("fluxmi: AutoEncoder.encode needs the GPU (libfluxmi has no CPU path)")
        if not self.encoder_loaded:
            raise RuntimeError("fluxmi: this autoencoder was loaded from a decoder-only checkpoint; encode() needs the encoder.* weights")
        e = self.encoder
        n_down = e.num_resolutions - 1
        if x.shape[-2] % (1 << n_down) or x.shape[-1] % (1 << n_down):
            raise ValueError(f"fluxmi: AutoEncoder.encode needs H and W to be multiples of {1 << n_down}")
        h = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()  # NHWC
        h = self._conv3(h, e.conv_in)
        for i_level in range(e.num_resolutions):
            for i_block in range(e.num_res_blocks):
                h = self._resnet(h, e.down[i_level].block[i_block])
            if i_level != e.num_resolutions - 1:
                h = self._conv3(h, e.down[i_level].downsample.conv, upsample=-2)
        h = self._resnet(h, e.mid.block_1)
        h = self._attn(h, e.mid.attn_1)
        h = self._resnet(h, e.mid.block_2)
        h = self._conv3(self._norm(h, e.norm_out, True), e.conv_out)
        return h.permute(0, 3, 1, 2).contiguous()

    @torch.inference_mode()
    def encode(self, x: Tensor, noise: Tensor | None = None) -> Tensor:
        """reference :326-329: z = scale_factor * (reg(encoder(x)) - shift_factor); `noise` replaces the randn_like draw (tests)."""
        z = self.reg(self.encode_moments(x), noise)
        return self.scale_factor * (z - self.shift_factor)

    def forward(self, x: Tensor) -> Tensor:
        return self.decode(self.encode(x))
