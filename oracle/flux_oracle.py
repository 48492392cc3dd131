"""CPU oracle for the Flux denoise hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file, and
only as the checker.  The product path (flux-fp8-api_amd/) never imports it and fails loudly when
libfluxmi.so is missing.

What this is: a functional, state-dict-driven restatement (torch CPU tensors, fp32/fp64 math where
the reference relies on torch internals) of the algorithm the reference executes on its hot path.
Every function cites the reference file:line it follows (paths relative to /root/reference).

Pinning status: the reference ships no tests / golden vectors for this path, so
the oracle is pinned against the reference ITSELF, imported unmodified in the build container by
oracle/gen_golden.py: that script asserts bit-equality between this file and the reference on seeded
inputs (F8Linear calibration trace, quantised bytes, every block, a multi-step Euler loop, LoRA fuse)
and writes the fixtures under tests/golden/.  tests/test_oracle_golden.py re-checks this file against
those fixtures everywhere (GPU box included, where /root/reference does not exist).

Third-party arithmetic: everything numeric in the reference happens inside PyTorch (unpinned,
requirements.txt has no torch line; README.md:130 asks for >= 2.4).  `torch._scaled_mm`
(float8_quantize.py:284) is restated by `scaled_mm_ref` below as  bf16( fp32(sum_k a*b) * (sa*sb) + bias );
the fp8 x fp8 products are exact in fp32, so only the accumulation order is implementation-defined.
`scaled_mm_fp64` is the order-independent version used for the <= 1 bf16-ulp GEMM gate.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

E4M3_MAX = 448.0  # torch.finfo(torch.float8_e4m3fn).max  (float8_quantize.py:52)
E5M2_MAX = 57344.0  # torch.finfo(torch.float8_e5m2).max  (float8_quantize.py:53)
NUM_SCALE_TRIALS = 12  # float8_quantize.py:42


# --------------------------------------------------------------------------------------
# fp8 scalar machinery                                      float8_quantize.py:195-246
# --------------------------------------------------------------------------------------
def amax_to_scale(amax: torch.Tensor, max_val: float) -> torch.Tensor:
    """float8_quantize.py:214-215: max_val / max(amax, 1e-12), itself capped at max_val."""
    return (max_val / torch.clamp(amax, min=1e-12)).clamp(max=max_val)


def to_fp8_saturated(x: torch.Tensor, scale: torch.Tensor, max_val: float) -> torch.Tensor:
    """float8_quantize.py:217-218.  `scale` is a 0-dim fp32 tensor, so `x * scale` stays in x.dtype
    (bf16 product, rounded) before the clamp.  The caller casts to fp8 (RNE)."""
    return (x * scale).clamp(-max_val, max_val)


def quantize_weight(w: torch.Tensor, f8=torch.float8_e4m3fn):
    """float8_quantize.py:195-207 -> (float8_data[N,K], scale, scale_reciprocal)."""
    max_val = torch.finfo(f8).max
    amax = torch.max(torch.abs(w)).float()
    scale = amax_to_scale(amax, max_val)
    data = to_fp8_saturated(w, scale, max_val).to(f8)
    return data, scale, scale.reciprocal()


def scaled_mm_ref(a8, w8, sa_recip, sb_recip, bias, out_dtype=torch.bfloat16, use_torch=True):
    """torch._scaled_mm as called at float8_quantize.py:284-292 (A[M,K] row-major, B = W[N,K].T)."""
    if use_torch and hasattr(torch, "_scaled_mm"):
        out = torch._scaled_mm(
            a8, w8.T, scale_a=sa_recip, scale_b=sb_recip, bias=bias, out_dtype=out_dtype,
            use_fast_accum=True,
        )
        return out[0] if isinstance(out, tuple) else out
    acc = a8.float() @ w8.float().T
    acc = acc * (sa_recip.float() * sb_recip.float())
    if bias is not None:
        acc = acc + bias.float()
    return acc.to(out_dtype)


def scaled_mm_fp64(a8, w8, sa_recip, sb_recip, bias):
    """Order-independent evaluation in fp64 (returned as fp64, not rounded)."""
    acc = a8.double() @ w8.double().T
    acc = acc * (sa_recip.double() * sb_recip.double())
    if bias is not None:
        acc = acc + bias.double()
    return acc


class F8LinearState:
    """State + forward of one reference F8Linear (float8_quantize.py:30-296), functional form."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor],
                 f8=torch.float8_e4m3fn, in_f8=torch.float8_e5m2):
        self.out_dtype = weight.dtype
        self.f8, self.in_f8 = f8, in_f8
        self.max_value = torch.finfo(f8).max
        self.input_max_value = torch.finfo(in_f8).max
        self.bias = bias
        self.float8_data, self.scale, self.scale_reciprocal = quantize_weight(weight, f8)
        self.input_amax_trials = torch.zeros(NUM_SCALE_TRIALS, dtype=torch.float32)
        self.trial_index = 0
        self.input_scale = None
        self.input_scale_reciprocal = None
        self.input_scale_initialized = False

    # float8_quantize.py:209-212
    def set_weight_tensor(self, w: torch.Tensor):
        self.float8_data, self.scale, self.scale_reciprocal = quantize_weight(w, self.f8)

    # lora_loading.py:615-631
    def dequantized_weight(self) -> torch.Tensor:
        return self.float8_data.float().mul(self.scale_reciprocal)

    def _q(self, x):
        return to_fp8_saturated(x, self.input_scale, self.input_max_value).to(self.in_f8)

    # float8_quantize.py:220-246 (+ the bypass at :273-276)
    def quantize_input(self, x: torch.Tensor) -> torch.Tensor:
        if self.input_scale_initialized:
            return self._q(x)
        if self.trial_index < NUM_SCALE_TRIALS:
            amax = torch.max(torch.abs(x)).float()
            self.input_amax_trials[self.trial_index] = amax
            self.trial_index += 1
            running = self.input_amax_trials[: self.trial_index].max()
            self.input_scale = amax_to_scale(running, self.input_max_value)
            self.input_scale_reciprocal = self.input_scale.reciprocal()
            return self._q(x)
        self.input_scale = amax_to_scale(self.input_amax_trials.max(), self.input_max_value)
        self.input_scale_reciprocal = self.input_scale.reciprocal()
        self.input_scale_initialized = True
        return self._q(x)

    # float8_quantize.py:272-296
    def __call__(self, x: torch.Tensor, trace: Optional[dict] = None, tag: str = "") -> torch.Tensor:
        x8 = self.quantize_input(x)
        lead = x8.shape[:-1]
        x8 = x8.reshape(-1, x8.shape[-1])
        if trace is not None:
            trace[tag + ".x8"] = x8
        out = scaled_mm_ref(x8, self.float8_data, self.input_scale_reciprocal,
                            self.scale_reciprocal, self.bias, self.out_dtype)
        if trace is not None:
            trace[tag + ".out"] = out
        return out.view(*lead, -1)


class PlainLinear:
    """nn.Linear left un-quantised (bf16 F.linear)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias

    def __call__(self, x, trace=None, tag=""):
        return F.linear(x, self.weight, self.bias)


# --------------------------------------------------------------------------------------
# model pieces                                                    modules/flux_model.py
# --------------------------------------------------------------------------------------
@dataclass
class FluxParams:  # mirrors the fields of modules/flux_model.py:24-36
    in_channels: int = 64
    vec_in_dim: int = 768
    context_in_dim: int = 4096
    hidden_size: int = 3072
    mlp_ratio: float = 4.0
    num_heads: int = 24
    depth: int = 19
    depth_single_blocks: int = 38
    axes_dim: List[int] = field(default_factory=lambda: [16, 56, 56])
    theta: int = 10_000
    qkv_bias: bool = True
    guidance_embed: bool = True


def rope_table(ids: torch.Tensor, axes_dim, theta, dtype) -> torch.Tensor:
    """modules/flux_model.py:49-57 + 82-92: [B,1,L,sum(axes)/2,2,2] in the flow dtype."""
    parts = []
    for i, d in enumerate(axes_dim):
        pos = ids[..., i]
        frac = torch.arange(0, d, 2, dtype=torch.float32) / d
        omega = 1.0 / (theta ** frac)
        ang = torch.einsum("...n,d->...nd", pos, omega)
        m = torch.stack([torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)], dim=-1)
        parts.append(m.reshape(*m.shape[:-1], 2, 2).type(dtype))
    return torch.cat(parts, dim=-3).unsqueeze(1)


def apply_rope(xq, xk, pe):
    """modules/flux_model.py:60-65 -- arithmetic stays in the flow dtype (no fp32 upcast)."""
    def rot(x):
        x_ = x.reshape(*x.shape[:-1], -1, 1, 2)
        return (pe[..., 0] * x_[..., 0] + pe[..., 1] * x_[..., 1]).reshape(*x.shape)
    return rot(xq), rot(xk)


def attention(q, k, v, pe):
    """modules/flux_model.py:41-45."""
    q, k = apply_rope(q, k, pe)
    x = F.scaled_dot_product_attention(q, k, v).transpose(1, 2)
    return x.reshape(*x.shape[:-2], -1)


def attention_exact(q, k, v, pe):
    """Same contract as `attention` with the softmax evaluated in fp64 and rounded once: an equally valid evaluation of
    flux_model.py:41-45 (SDPA's internal order / precision is implementation-defined).  The tests use it to measure how far a block's
    output moves under rounding-level changes of the attention output -- the noise floor the HIP engine is gated against."""
    q, k = apply_rope(q, k, pe)
    x = attention_fp64(q, k, v).to(q.dtype).transpose(1, 2)
    return x.reshape(*x.shape[:-2], -1)


def attention_fp64(q, k, v):
    """softmax(q k^T / sqrt(d)) v in fp64 on already-rotated bf16 q,k (independent check)."""
    s = (q.double() @ k.double().transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.softmax(s, dim=-1) @ v.double()


def rms_norm(x, scale):
    """modules/flux_model.py:158-164: fp32 rms_norm over the head dim, eps 1e-6, cast back."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * scale.float()).to(x.dtype)


def timestep_embedding(t: torch.Tensor, dim=256, max_period=10000, time_factor=1000.0):
    """modules/flux_model.py:95-116.  NB `time_factor * t` is evaluated in t.dtype (bf16)."""
    t = time_factor * t
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def layer_norm(x):
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6) (flux_model.py:282,290,316,324,453,491)."""
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def split_heads(x, num_heads):
    """flux_model.py:351-354 / 476-477: [B,L,3*H*D] -> q,k,v each [B,H,L,D]."""
    B, L, D3 = x.shape
    return x.reshape(B, L, 3, num_heads, D3 // (3 * num_heads)).permute(2, 0, 3, 1, 4)


class FluxOracle:
    """Functional Flux (modules/flux_model.py:506-716) over a BFL-format state dict.

    quantize: None (bf16 nn.Linear everywhere -- the reference "bf16 flow path"), or a dict
      {"modulation": bool, "embedders": bool} selecting which layers become F8Linear exactly as
      quantize_flow_transformer_and_dispatch_float8 does (float8_quantize.py:395-496):
      every Linear inside double/single blocks (Modulation only if modulation=True), the five
      embedder modules only if embedders=True, final_layer never.
    """

    def __init__(self, sd: Dict[str, torch.Tensor], params: FluxParams, dtype=torch.bfloat16,
                 quantize: Optional[dict] = None):
        self.p = params
        self.dtype = dtype
        self.sd = sd
        self.quantize = quantize
        self.lin: Dict[str, object] = {}
        for key in sd:
            if not key.endswith(".weight"):
                continue
            name = key[: -len(".weight")]
            w = sd[key]
            if w.ndim != 2:
                continue
            b = sd.get(name + ".bias")
            self.lin[name] = F8LinearState(w, b) if self._is_quantized(name) else PlainLinear(w, b)

    def _is_quantized(self, name: str) -> bool:
        if self.quantize is None:
            return False
        if name.startswith("final_layer"):
            return False
        if name.startswith(("double_blocks", "single_blocks")):
            if ".img_mod." in name or ".txt_mod." in name or ".modulation." in name:
                return bool(self.quantize.get("modulation", True))
            return True
        return bool(self.quantize.get("embedders", False))

    def n_f8(self):
        return sum(isinstance(v, F8LinearState) for v in self.lin.values())

    def freeze_input_scales(self):
        """Stop calibrating NOW: what the reference does after its 13th call (float8_quantize.py:239-246), forced early by
        setting `input_scale_initialized` with the running scale (tests at full geometry cannot afford 13 CPU calls)."""
        for m in self.lin.values():
            if isinstance(m, F8LinearState):
                assert m.input_scale is not None, "freeze_input_scales before the first call"
                m.input_scale_initialized = True

    # ---- sub-modules --------------------------------------------------------------------
    def _mlp_embedder(self, prefix, x, trace=None):
        """flux_model.py:154-155."""
        h = self.lin[prefix + ".in_layer"](x, trace, prefix + ".in_layer")
        return self.lin[prefix + ".out_layer"](F.silu(h), trace, prefix + ".out_layer")

    def _modulation(self, prefix, vec, n, trace=None):
        """flux_model.py:251-257 -> list of n tensors [B,1,H]."""
        out = self.lin[prefix + ".lin"](F.silu(vec), trace, prefix + ".lin")
        return out[:, None, :].chunk(n, dim=-1)

    def double_block(self, i, img, txt, vec, pe, trace=None, attn_fn=None):
        """flux_model.py:356-400 (bf16 flow: no clamp).  attn_fn: alternative evaluation of `attention` (tests' noise probe)."""
        pre = f"double_blocks.{i}"
        H = self.p.num_heads
        im = self._modulation(pre + ".img_mod", vec, 6, trace)
        tm = self._modulation(pre + ".txt_mod", vec, 6, trace)
        img_modulated = (1 + im[1]) * layer_norm(img) + im[0]
        img_qkv = self.lin[pre + ".img_attn.qkv"](img_modulated, trace, pre + ".img_attn.qkv")
        iq, ik, iv = split_heads(img_qkv, H)
        iq = rms_norm(iq, self.sd[pre + ".img_attn.norm.query_norm.scale"])
        ik = rms_norm(ik, self.sd[pre + ".img_attn.norm.key_norm.scale"])
        txt_modulated = (1 + tm[1]) * layer_norm(txt) + tm[0]
        txt_qkv = self.lin[pre + ".txt_attn.qkv"](txt_modulated, trace, pre + ".txt_attn.qkv")
        tq, tk, tv = split_heads(txt_qkv, H)
        tq = rms_norm(tq, self.sd[pre + ".txt_attn.norm.query_norm.scale"])
        tk = rms_norm(tk, self.sd[pre + ".txt_attn.norm.key_norm.scale"])
        q = torch.cat((tq, iq), dim=2)
        k = torch.cat((tk, ik), dim=2)
        v = torch.cat((tv, iv), dim=2)
        attn = (attn_fn or attention)(q, k, v, pe)
        Lt = txt.shape[1]
        t_attn, i_attn = attn[:, :Lt], attn[:, Lt:]
        if trace is not None:
            trace[pre + ".img_modulated"] = img_modulated
            trace[pre + ".txt_modulated"] = txt_modulated
            trace[pre + ".img_qkv"] = img_qkv
            trace[pre + ".txt_qkv"] = txt_qkv
            qr, kr = apply_rope(q, k, pe)
            trace[pre + ".q_rot"], trace[pre + ".k_rot"], trace[pre + ".v"] = qr, kr, v
            trace[pre + ".attn"] = attn
        img = img + im[2] * self.lin[pre + ".img_attn.proj"](i_attn, trace, pre + ".img_attn.proj")
        if trace is not None:
            trace[pre + ".img_mid"] = img
        h = self.lin[pre + ".img_mlp.0"]((1 + im[4]) * layer_norm(img) + im[3], trace, pre + ".img_mlp.0")
        img = img + im[5] * self.lin[pre + ".img_mlp.2"](F.gelu(h, approximate="tanh"), trace, pre + ".img_mlp.2")
        txt = txt + tm[2] * self.lin[pre + ".txt_attn.proj"](t_attn, trace, pre + ".txt_attn.proj")
        if trace is not None:
            trace[pre + ".txt_mid"] = txt
        h = self.lin[pre + ".txt_mlp.0"]((1 + tm[4]) * layer_norm(txt) + tm[3], trace, pre + ".txt_mlp.0")
        txt = txt + tm[5] * self.lin[pre + ".txt_mlp.2"](F.gelu(h, approximate="tanh"), trace, pre + ".txt_mlp.2")
        return img, txt

    def single_block(self, i, x, vec, pe, trace=None, attn_fn=None):
        """flux_model.py:467-485."""
        pre = f"single_blocks.{i}"
        Hd = self.p.hidden_size
        shift, scale, gate = self._modulation(pre + ".modulation", vec, 3, trace)
        x_mod = (1 + scale) * layer_norm(x) + shift
        lin1 = self.lin[pre + ".linear1"](x_mod, trace, pre + ".linear1")
        qkv, mlp = torch.split(lin1, [3 * Hd, lin1.shape[-1] - 3 * Hd], dim=-1)
        q, k, v = split_heads(qkv, self.p.num_heads)
        q = rms_norm(q, self.sd[pre + ".norm.query_norm.scale"])
        k = rms_norm(k, self.sd[pre + ".norm.key_norm.scale"])
        attn = (attn_fn or attention)(q, k, v, pe)
        cat = torch.cat((attn, F.gelu(mlp, approximate="tanh")), 2)
        out = self.lin[pre + ".linear2"](cat, trace, pre + ".linear2")
        if trace is not None:
            trace[pre + ".x_mod"], trace[pre + ".lin1"] = x_mod, lin1
            trace[pre + ".attn"], trace[pre + ".cat"] = attn, cat
        return x + gate * out

    def final_layer(self, x, vec):
        """flux_model.py:499-503 (shift first, then scale)."""
        mod = self.lin["final_layer.adaLN_modulation.1"](F.silu(vec))
        shift, scale = mod.chunk(2, dim=1)
        x = (1 + scale[:, None, :]) * layer_norm(x) + shift[:, None, :]
        return self.lin["final_layer.linear"](x)

    def embed_vec(self, timesteps, y, guidance, trace=None):
        """flux_model.py:687-697."""
        vec = self._mlp_embedder("time_in", timestep_embedding(timesteps, 256).type(self.dtype), trace)
        if self.p.guidance_embed:
            if guidance is None:
                raise ValueError("Didn't get guidance strength for guidance distilled model.")
            vec = vec + self._mlp_embedder("guidance_in", timestep_embedding(guidance, 256).type(self.dtype), trace)
        return vec + self._mlp_embedder("vector_in", y, trace)

    def forward(self, img, img_ids, txt, txt_ids, timesteps, y, guidance=None, trace=None):
        """flux_model.py:672-716."""
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        img = self.lin["img_in"](img, trace, "img_in")
        vec = self.embed_vec(timesteps, y, guidance, trace)
        txt = self.lin["txt_in"](txt, trace, "txt_in")
        pe = rope_table(torch.cat((txt_ids, img_ids), dim=1), self.p.axes_dim, self.p.theta, self.dtype)
        if trace is not None:
            trace["vec"], trace["pe"], trace["img_in.out"], trace["txt_in.out"] = vec, pe, img, txt
        for i in range(self.p.depth):
            img, txt = self.double_block(i, img, txt, vec, pe, trace)
            if trace is not None:
                trace[f"double_blocks.{i}.img_out"], trace[f"double_blocks.{i}.txt_out"] = img, txt
        x = torch.cat((txt, img), 1)
        for i in range(self.p.depth_single_blocks):
            x = self.single_block(i, x, vec, pe, trace)
            if trace is not None:
                trace[f"single_blocks.{i}.out"] = x
        x = x[:, txt.shape[1]:, ...]
        return self.final_layer(x, vec)

    # ---- LoRA fuse (config 5)                         lora_loading.py:509-577, 615-631, 678-689
    def fuse_lora(self, lora: Dict[str, torch.Tensor], lora_scale: float = 1.0, sign: float = 1.0):
        names = sorted({k.split(".lora_")[0].replace(".alpha", "") for k in lora})
        for name in names:
            A, Bm = lora.get(name + ".lora_A.weight"), lora.get(name + ".lora_B.weight")
            if A is None or Bm is None:
                continue
            mod = self.lin[name]
            delta = lora_delta(A, Bm, lora.get(name + ".alpha"), lora_scale)
            if isinstance(mod, F8LinearState):
                w = mod.dequantized_weight() + sign * delta
                mod.set_weight_tensor(w.type(mod.out_dtype))
            else:
                mod.weight = (mod.weight.float() + sign * delta).type(mod.weight.dtype)


def lora_delta(lora_A, lora_B, alpha, lora_scale):
    """lora_loading.py:509-544 (incl. the 'uneven rank' chunk-sum for fused qkv)."""
    rank = lora_B.shape[1]
    if alpha is None:
        alpha = rank
    a = lora_A.float()
    b = lora_B.float()
    if alpha != rank:
        a = a * alpha / rank
    if lora_B.shape[1] != lora_A.shape[0]:
        n = int(lora_A.shape[0] / lora_B.shape[1])
        out = torch.zeros((lora_B.shape[0], lora_A.shape[1]), dtype=torch.float32)
        for chunk in a.chunk(n, dim=0):
            out = out + (lora_scale * torch.mm(b, chunk))
        return out
    return lora_scale * torch.mm(b, a)


# --------------------------------------------------------------------------------------
# pipeline-side pieces of the path                                      flux_pipeline.py
# --------------------------------------------------------------------------------------
def get_schedule(num_steps: int, image_seq_len: int, base_shift=0.5, max_shift=1.15, shift=True):
    """flux_pipeline.py:314-344."""
    ts = torch.linspace(1, 0, num_steps + 1)
    if shift:
        m = (max_shift - base_shift) / (4096 - 256)
        mu = m * image_seq_len + (base_shift - m * 256)
        ts = math.exp(mu) / (math.exp(mu) + (1 / ts - 1) ** 1.0)
    return ts.tolist()


def pack_latent(x: torch.Tensor) -> torch.Tensor:
    """flux_pipeline.py:267-271: [B,C,h,w] -> [B,(h/2)(w/2),C*4] with channel order (c,ph,pw)."""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latent(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    """flux_pipeline.py:440-448."""
    b = x.shape[0]
    h, w = math.ceil(height / 16), math.ceil(width / 16)
    x = x.reshape(b, h, w, -1, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(b, -1, h * 2, w * 2)


def make_ids(batch, h2, w2, txt_len, dtype):
    """flux_pipeline.py:280-292 (img ids in the flow dtype) and flux_emphasis.py:433-439 (txt ids = 0)."""
    img_ids = torch.zeros(h2, w2, 3, dtype=dtype)
    img_ids[..., 1] = img_ids[..., 1] + torch.arange(h2, dtype=dtype)[:, None]
    img_ids[..., 2] = img_ids[..., 2] + torch.arange(w2, dtype=dtype)[None, :]
    img_ids = img_ids[None].repeat(batch, 1, 1, 1).flatten(1, 2)
    txt_ids = torch.zeros(batch, txt_len, 3, dtype=dtype)
    return img_ids, txt_ids


def denoise(model: FluxOracle, img, img_ids, txt, txt_ids, vec, timesteps: List[float],
            guidance: float = 3.5, collect=None):
    """flux_pipeline.py:619-651: Euler loop; t and guidance are created in the flow dtype."""
    dtype = model.dtype
    g = torch.full((img.shape[0],), guidance, dtype=dtype)
    for t_curr, t_prev in zip(timesteps[:-1], timesteps[1:]):
        t_vec = torch.full((img.shape[0],), t_curr, dtype=dtype)
        pred = model.forward(img=img, img_ids=img_ids, txt=txt, txt_ids=txt_ids, y=vec,
                             timesteps=t_vec, guidance=g)
        img = img + (t_prev - t_curr) * pred
        if collect is not None:
            collect.append(img)
    return img
