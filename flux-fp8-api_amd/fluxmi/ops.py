"""Operator-level host wrappers: torch CUDA(HIP) tensors in, libfluxmi kernels underneath.

PyTorch is plumbing only (device memory + current stream).  Every function launches on
`torch.cuda.current_stream()` and never synchronises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import E4M3, E5M2, GemmGroup, call

F8_DTYPES = {torch.float8_e4m3fn: E4M3, torch.float8_e5m2: E5M2}
F8_MAX = {E4M3: 448.0, E5M2: 57344.0}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"fluxmi: {name} must live on the GPU (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"fluxmi: {name} must be {dtype}, got {t.dtype}")
    return t


def fmt_of(dtype) -> int:
    return F8_DTYPES[dtype]


def dtype_of(fmt: int):
    return torch.float8_e5m2 if fmt == E5M2 else torch.float8_e4m3fn


# ---- quantisation ------------------------------------------------------------------------------
def quantize_act(x: torch.Tensor, scale: torch.Tensor, fmt: int = E5M2, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float8_quantize.py:217-218 + cast.  x: bf16 [..., K] (last dim contiguous)."""
    _req(x, torch.bfloat16, "x")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    if out is None:
        out = torch.empty(x2.shape, dtype=dtype_of(fmt), device=x.device)
    call("fluxmi_quantize_act", _p(x2), _p(out), _p(scale), x2.shape[0], x2.shape[1], x2.stride(0), out.stride(0), fmt, _stream())
    return out.view(*x.shape[:-1], x.shape[-1])


def amax(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, torch.bfloat16, "x")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    if out is None:
        out = torch.zeros((), dtype=torch.float32, device=x.device)
    call("fluxmi_amax", _p(x2), _p(out), x2.shape[0], x2.shape[1], x2.stride(0), _stream())
    return out


def calib_update(amax_t, trials, scale, scale_recip, trial_index: int, num_trials: int, max_val: float) -> None:
    call("fluxmi_calib_update", _p(amax_t), _p(trials), _p(scale), _p(scale_recip), trial_index, num_trials, max_val, _stream())


def quantize_weight(w: torch.Tensor, fmt: int = E4M3):
    """F8Linear.quantize_weight (float8_quantize.py:195-207) -> (float8_data, scale, scale_reciprocal)."""
    _req(w, torch.bfloat16, "weight")
    w = w.contiguous()
    q = torch.empty(w.shape, dtype=dtype_of(fmt), device=w.device)
    tmp = torch.zeros(3, dtype=torch.float32, device=w.device)
    call("fluxmi_quantize_weight", _p(w), _p(q), _p(tmp[0:1]), _p(tmp[1:2]), _p(tmp[2:3]), w.shape[0], w.shape[1], fmt, _stream())
    return q, tmp[1].clone(), tmp[2].clone()


def dequant(q: torch.Tensor, scale_recip: torch.Tensor) -> torch.Tensor:
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    call("fluxmi_dequant", _p(q), _p(out), _p(scale_recip), q.numel(), fmt_of(q.dtype), _stream())
    return out


def lora_fuse_f8(w8: torch.Tensor, w_scale: torch.Tensor, w_scale_recip: torch.Tensor, lora_B: torch.Tensor,
                 lora_A: torch.Tensor, lora_scale: float, n_chunks: int = 1) -> None:
    """In-place LoRA fuse + re-quantise of an e4m3 weight (lora_loading.py:509-577,615-631,686-687)."""
    N, K = w8.shape
    R = lora_B.shape[1]
    work = torch.empty(2 * N * K, dtype=torch.float32, device=w8.device)
    tmp = torch.zeros(1, dtype=torch.float32, device=w8.device)
    if w8.dtype != torch.float8_e4m3fn:
        raise TypeError(f"fluxmi: LoRA fuse needs float8_e4m3fn weights (the GEMM kernels fix the weight operand format), got {w8.dtype}")
    call("fluxmi_lora_fuse_f8", _p(w8), _p(w_scale), _p(w_scale_recip), _p(lora_B.float().contiguous()),
         _p(lora_A.float().contiguous()), N, K, R, n_chunks, float(lora_scale), _p(work), _p(tmp), _stream())


# ---- linear ----------------------------------------------------------------------------------------
def make_group(A, W, bias, sa_recip, sb_recip, Cout, M, lda, ldc, *, C2=None, ldc2=0, gate=None, resid=None, ldr=0,
               q_scale=None, split_n=0, c2_col0=0, vt_out=None, vt_ld=0, tok0=0, vt_rows=0, kv_col0=0, heads=0, k_out=None, pe=None,
               k_norm=None, k_rows=0, q_lut=None, k_f16=False, W_pairs=None, a_pairs=False, c8_pairs=False) -> GemmGroup:
    g = GemmGroup()
    g.q_lut = q_lut
    g.W_pairs = W_pairs
    g.a_pairs, g.c8_pairs = int(a_pairs), int(c8_pairs)
    g.k_f16 = int(k_f16)
    g.vt_out, g.k_out, g.pe, g.k_norm = vt_out, k_out, pe, k_norm
    g.vt_ld, g.k_rows, g.tok0, g.vt_rows, g.kv_col0, g.heads = vt_ld, k_rows, tok0, vt_rows, kv_col0, heads
    g.A, g.W, g.bias, g.sa_recip, g.sb_recip = A, W, bias, sa_recip, sb_recip
    g.C, g.C2, g.gate, g.resid, g.q_scale = Cout, C2, gate, resid, q_scale
    g.lda, g.ldc, g.ldc2, g.ldr, g.M, g.split_n, g.c2_col0 = lda, ldc, ldc2, ldr, M, split_n, c2_col0
    return g


def gemm_grouped(groups: Sequence[GemmGroup], N: int, K: int, is_fp8: bool, act_fmt: int, epilogue: int, tile_cfg: int = -1) -> None:
    arr = (GemmGroup * len(groups))(*groups)
    call("fluxmi_gemm_grouped", arr, len(groups), N, K, int(is_fp8), act_fmt, epilogue, tile_cfg, _stream())


def pair_rows(w: torch.Tensor) -> torch.Tensor:
    """[R, C] (1-byte or 2-byte elements, row bytes % 64 == 0, R even) -> the same bytes in the row-pair layout [R/2][row_bytes/64][2][64]
    (fluxmi_gemm_group_t.W_pairs), as a tensor of w's shape and dtype."""
    assert w.dim() == 2 and w.is_contiguous()
    out = torch.empty_like(w)
    call("fluxmi_pair_rows", _p(w), _p(out), w.shape[0], w.shape[1] * w.element_size(), _stream())
    return out


def build_quant_lut(scale: torch.Tensor, fmt: int = E5M2, act: int = 1) -> torch.Tensor:
    """64 KiB table bf16 bit pattern -> fp8 byte of quantise(act(x)) (act: 0 none, 1 gelu-tanh, 2 silu) for a frozen input scale."""
    lut = torch.empty(65536, dtype=torch.uint8, device=scale.device)
    call("fluxmi_build_quant_lut", _p(scale), fmt, act, _p(lut), _stream())
    return lut


def linear(x, W, bias=None, sa_recip=None, sb_recip=None, *, epilogue=_lib.EPI_BF16, gate=None, resid=None, q_scale=None,
           out=None, out2=None, split_n=0, c2_col0=0, tile_cfg=-1, out_fmt=E5M2, q_lut=None):
    """One (F8)Linear: x [M,K] fp8 or bf16, W [N,K].  Returns the primary output tensor."""
    is_fp8 = W.dtype in F8_DTYPES
    M, K = x.shape
    N = W.shape[0]
    act_fmt = fmt_of(x.dtype) if is_fp8 else out_fmt
    fp8_out = epilogue in (_lib.EPI_GELU_QUANT, _lib.EPI_QUANT, _lib.EPI_SILU_QUANT)
    n_primary = split_n if epilogue == _lib.EPI_SPLIT else N
    if out is None:
        out = torch.empty((M, n_primary), dtype=dtype_of(act_fmt) if fp8_out else torch.bfloat16, device=x.device)
    g = make_group(_p(x), _p(W), _p(bias), _p(sa_recip), _p(sb_recip), _p(out), M, x.stride(0), out.stride(0),
                   C2=_p(out2), ldc2=out2.stride(0) if out2 is not None else 0, gate=_p(gate), resid=_p(resid),
                   ldr=resid.stride(0) if resid is not None else 0, q_scale=_p(q_scale), split_n=split_n, c2_col0=c2_col0, q_lut=_p(q_lut))
    gemm_grouped([g], N, K, is_fp8, act_fmt, epilogue, tile_cfg)
    return out


def gemv(x, W, bias=None, in_scale=None, sa_recip=None, sb_recip=None, pre_silu=False, act_fmt=E5M2, out=None):
    B, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((B, N), dtype=torch.bfloat16, device=x.device)
    call("fluxmi_gemv", _p(x), x.stride(0), _p(W), _p(bias), _p(in_scale), _p(sa_recip), _p(sb_recip), _p(out), out.stride(0),
         B, N, K, int(W.dtype in F8_DTYPES), act_fmt, int(pre_silu), _stream())
    return out


# ---- block elementwise ---------------------------------------------------------------------------------
def ln_modulate(x, shift0, scale0, shift1=None, scale1=None, split=None, q_scale0=None, q_scale1=None, fmt=E5M2):
    """x [B,L,H] bf16; shift/scale [B,H] (row stride = mod_bstride).  fp8 output iff q_scale0 is given."""
    B, L, H = x.shape
    split = L if split is None else split
    shift1 = shift0 if shift1 is None else shift1
    scale1 = scale0 if scale1 is None else scale1
    out_fp8 = q_scale0 is not None
    q_scale1 = q_scale0 if q_scale1 is None else q_scale1
    out = torch.empty((B, L, H), dtype=dtype_of(fmt) if out_fp8 else torch.bfloat16, device=x.device)
    call("fluxmi_ln_modulate", _p(x), x.stride(1), _p(out), H, _p(shift0), _p(scale0), _p(shift1), _p(scale1), shift0.stride(0),
         _p(q_scale0), _p(q_scale1), B, L, split, H, int(out_fp8), fmt, _stream())
    return out


def act(x, mode: int):
    x2 = x.reshape(-1, x.shape[-1])
    y = torch.empty_like(x2)
    call("fluxmi_act", _p(x2), _p(y), x2.shape[0], x2.shape[1], x2.stride(0), y.stride(0), mode, _stream())
    return y.view(x.shape)


def gate_residual(x, y, gate):
    B, L, H = x.shape
    out = torch.empty_like(x)
    call("fluxmi_gate_residual", _p(x), _p(y), _p(gate), _p(out), B, L, H, H, H, H, gate.stride(0), _stream())
    return out


def add(a, b):
    z = torch.empty_like(a)
    call("fluxmi_add", _p(a), _p(b), _p(z), a.numel(), _stream())
    return z


# ---- attention path -------------------------------------------------------------------------------------
def rope_tables_host(axes_dim, theta):
    """Host-side constants, evaluated with the reference's own torch expressions (flux_model.py:50-51)."""
    omega, axis = [], []
    for i, d in enumerate(axes_dim):
        frac = torch.arange(0, d, 2, dtype=torch.float32) / d
        om = 1.0 / (theta ** frac)
        omega.append(om)
        axis += [i] * om.numel()
    return torch.cat(omega).contiguous(), torch.tensor(axis, dtype=torch.int32)


def timestep_freqs_host(half=128, max_period=10000):
    import math
    return torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)


def rope_table(ids: torch.Tensor, axes_dim, theta) -> torch.Tensor:
    """ids bf16 [B,L,3] -> pe bf16 [B,L,64,2] (cos, sin)."""
    om, ax = rope_tables_host(axes_dim, theta)
    om, ax = om.to(ids.device), ax.to(ids.device)
    B, L, n_axes = ids.shape
    pe = torch.empty((B, L, om.numel(), 2), dtype=torch.bfloat16, device=ids.device)
    call("fluxmi_rope_table", _p(ids.contiguous()), _p(om), _p(ax), _p(pe), B * L, n_axes, om.numel(), _stream())
    return pe


def qkv_rope(qkv, pe, q_scale0, k_scale0, q_scale1=None, k_scale1=None, split=None, heads=None, skip_q=False, k_f16=False):
    """qkv bf16 [B,L,>=3*H*128] -> Q,K [B,H,L,128], VT [B,H,128,Lp].  skip_q: only K and VT (Q is returned as None).
    k_f16: K is returned as float16 (the operand format of the attention kernel's folded schedule; pass it on to attention*)."""
    B, L, _ = qkv.shape
    split = L if split is None else split
    q_scale1 = q_scale0 if q_scale1 is None else q_scale1
    k_scale1 = k_scale0 if k_scale1 is None else k_scale1
    Lp = (L + 63) // 64 * 64
    K = torch.empty((B, heads, L, 128), dtype=torch.float16 if k_f16 else torch.bfloat16, device=qkv.device)
    Q = None if skip_q else torch.empty((B, heads, L, 128), dtype=torch.bfloat16, device=qkv.device)
    VT = torch.empty((B, heads, 128, Lp), dtype=torch.bfloat16, device=qkv.device)
    call("fluxmi_qkv_rope", _p(qkv), qkv.stride(1), _p(pe), _p(q_scale0), _p(k_scale0), _p(q_scale1), _p(k_scale1), _p(Q), _p(K),
         _p(VT), B, L, Lp, heads, split, int(k_f16), _stream())
    return Q, K, VT


def attention(Q, K, VT, q_scale0=None, q_scale1=None, split=None, fmt=E5M2, out=None, col_off=0):
    """K may be bfloat16 or float16 (float16 selects the folded kernel, see include/fluxmi.h)."""
    B, H, L, _ = Q.shape
    Lp = VT.shape[-1]
    split = L if split is None else split
    out_fp8 = q_scale0 is not None
    q_scale1 = q_scale0 if q_scale1 is None else q_scale1
    if out is None:
        out = torch.empty((B, L, H * 128), dtype=dtype_of(fmt) if out_fp8 else torch.bfloat16, device=Q.device)
    call("fluxmi_attention", _p(Q), _p(K), _p(VT), _p(out), out.stride(1), col_off, int(out_fp8), _p(q_scale0), _p(q_scale1), split,
         B, L, Lp, H, fmt, int(K.dtype == torch.float16), _stream())
    return out


def unpair_rows(w: torch.Tensor) -> torch.Tensor:
    """inverse of pair_rows: the row-pair layout [R/2][row_bytes/64][2][64] back to plain rows [R, C]"""
    assert w.dim() == 2 and w.is_contiguous()
    out = torch.empty_like(w)
    call("fluxmi_unpair_rows", _p(w), _p(out), w.shape[0], w.shape[1] * w.element_size(), _stream())
    return out


def attention_plan(B: int, L: int, H: int):
    """fluxmi_attention_plan: None when an attention launch of this shape runs one workgroup per task, else a dict with the balanced
    grid's geometry: n_per_x / full_per_x (tasks per XCD, of which whole), `thin` (the default takes the plan) and `pieces` in launch order (dicts with tloc, pidx, np,
    base, tb, len).  Host arithmetic only -- works without a GPU."""
    import ctypes as C

    n, f, npc = C.c_int(), C.c_int(), C.c_int()
    raw = (C.c_ulonglong * 64)()
    kind = _lib.lib.fluxmi_attention_plan(B, L, H, C.byref(n), C.byref(f), C.byref(npc), raw)
    if not kind:
        return None
    pieces = [dict(tloc=v & 255, pidx=(v >> 8) & 255, np=(v >> 16) & 255, base=(v >> 24) & 255, tb=(v >> 32) & 0xFFFF, len=(v >> 48) & 0xFFFF)
              for v in list(raw)[: npc.value]]
    return dict(n_per_x=n.value, full_per_x=f.value, pieces=pieces, thin=kind == 1)  # thin: what fluxmi_tuning_t.attn_split = 1 takes


def attention_rawq(qkv, pe, qn_scale0, K, VT, qn_scale1=None, q_scale0=None, q_scale1=None, split=None, fmt=E5M2, out=None, col_off=0):
    """Attention with Q read raw from the qkv GEMM output [B,L,>=H*128] (QKNorm + RoPE applied on load); K, VT from qkv_rope."""
    B, H, L, _ = K.shape
    Lp = VT.shape[-1]
    split = L if split is None else split
    qn_scale1 = qn_scale0 if qn_scale1 is None else qn_scale1
    out_fp8 = q_scale0 is not None
    q_scale1 = q_scale0 if q_scale1 is None else q_scale1
    if out is None:
        out = torch.empty((B, L, H * 128), dtype=dtype_of(fmt) if out_fp8 else torch.bfloat16, device=K.device)
    call("fluxmi_attention_rawq", _p(qkv), qkv.stride(1), _p(pe), _p(qn_scale0), _p(qn_scale1), _p(K), _p(VT), _p(out), out.stride(1),
         col_off, int(out_fp8), _p(q_scale0), _p(q_scale1), split, B, L, Lp, H, fmt, int(K.dtype == torch.float16), _stream())
    return out


def timestep_embedding(t: torch.Tensor, freqs: torch.Tensor, time_factor: float = 1000.0) -> torch.Tensor:
    B = t.shape[0]
    half = freqs.numel()
    out = torch.empty((B, 2 * half), dtype=torch.bfloat16, device=t.device)
    call("fluxmi_timestep_embedding", _p(t), _p(freqs), _p(out), B, half, time_factor, _stream())
    return out


def euler_(img: torch.Tensor, pred: torch.Tensor, dt: float) -> torch.Tensor:
    dts = torch.tensor([dt], dtype=torch.float32, device=img.device)
    call("fluxmi_euler", _p(img), _p(pred), _p(dts), None, img.numel(), _stream())
    return img


# ---- VAE decoder pieces (NHWC bf16) -------------------------------------------------------------------------------------
def conv3x3_ok(cin: int, cout: int) -> bool:
    """shapes fluxmi_conv3x3 (the implicit-GEMM 3x3 convolution) takes: 64-channel K-steps, 128-column tiles"""
    return cin % 64 == 0 and cout % 128 == 0


def conv3x3(x: torch.Tensor, w2: torch.Tensor, bias=None, upsample: int = 1, resid=None, gate=None) -> torch.Tensor:
    """x [B, Hin, Win, C] bf16 (NHWC), w2 [Cout, 9*C] bf16 with K ordered (dy, dx, c) -> [B, H, W, Cout]: the 3x3 convolution as an IMPLICIT
    GEMM (no patch matrix).  `upsample` as in im2col3x3; resid [B, H, W, Cout] (+ gate [Cout], default ones): out = resid + gate * y."""
    _req(x, torch.bfloat16, "x")
    _req(w2, torch.bfloat16, "w2")
    x = x.contiguous()
    B, Hi, Wi, Cc = x.shape
    Cout = w2.shape[0]
    if w2.shape[1] != 9 * Cc or not conv3x3_ok(Cc, Cout):
        raise ValueError(f"conv3x3: needs w2 [Cout, 9*C], C % 64 == 0, Cout % 128 == 0 (C={Cc}, w2={tuple(w2.shape)})")
    if upsample == -2:
        if Hi % 2 or Wi % 2:
            raise ValueError("conv3x3: the stride-2 mode needs even input dims")
        H, W = Hi // 2, Wi // 2
    else:
        H, W = Hi * upsample, Wi * upsample
    out = torch.empty((B, H, W, Cout), dtype=torch.bfloat16, device=x.device)
    if resid is not None:
        _req(resid, torch.bfloat16, "resid")
        resid = resid.contiguous()
        if gate is None:
            gate = torch.ones(Cout, dtype=torch.bfloat16, device=x.device)
    call("fluxmi_conv3x3", _p(x), _p(w2.contiguous()), _p(bias), _p(gate), _p(resid), _p(out), B, H, W, Cc, Cout, upsample, _stream())
    return out


def im2col3x3(x: torch.Tensor, upsample: int = 1) -> torch.Tensor:
    """x [B, Hin, Win, C] bf16 -> patch matrix [B*H*W, 9*C], column order (dy, dx, c).  upsample = 1: H = Hin; 2: nearest 2x upsample
    first (H = 2*Hin); -2: stride-2 window with zero pad on the right/bottom only (H = Hin/2, the reference's Downsample)."""
    _req(x, torch.bfloat16, "x")
    x = x.contiguous()
    B, Hi, Wi, Cc = x.shape
    if upsample == -2:
        if Hi % 2 or Wi % 2:
            raise ValueError("im2col3x3: the stride-2 mode needs even input dims")
        H, W = Hi // 2, Wi // 2
    else:
        H, W = Hi * upsample, Wi * upsample
    col = torch.empty((B * H * W, 9 * Cc), dtype=torch.bfloat16, device=x.device)
    call("fluxmi_im2col3x3", _p(x), _p(col), B, H, W, Cc, upsample, _stream())
    return col


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, swish: bool = True, eps: float = 1e-6) -> torch.Tensor:
    """x [B, P, C] bf16 (P pixels), 32 groups; fp32 statistics, one bf16 rounding at the end."""
    _req(x, torch.bfloat16, "x")
    x = x.contiguous()
    B, P, Cc = x.shape
    work = torch.empty((B * ((P + 511) // 512) + B) * 64, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    call("fluxmi_groupnorm", _p(x), _p(gamma), _p(beta), _p(y), _p(work), B, P, Cc, int(swish), float(eps), _stream())
    return y


def softmax_rows(S: torch.Tensor, scale: float) -> torch.Tensor:
    _req(S, torch.bfloat16, "S")
    S = S.contiguous()
    P = torch.empty_like(S)
    call("fluxmi_softmax_rows", _p(S), _p(P), S.shape[0], S.shape[1], S.stride(0), float(scale), _stream())
    return P


# ---- text-conditioning encoder pieces (bf16) ------------------------------------------------------------------------------
def row_norm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, eps: float = 1e-6, rms: bool = True) -> torch.Tensor:
    """x [rows, D] bf16.  rms=True: T5LayerNorm (no mean / bias, weight applied after a bf16 rounding); else LayerNorm."""
    _req(x, torch.bfloat16, "x")
    if x.stride(-1) != 1:
        x = x.contiguous()
    y = torch.empty((x.shape[0], x.shape[1]), dtype=torch.bfloat16, device=x.device)
    call("fluxmi_row_norm", _p(x), _p(weight), _p(bias) if bias is not None else None, _p(y), x.shape[0], x.shape[1], x.stride(0), y.stride(0),
         float(eps), 0 if rms else 1, _stream())
    return y


def act_mul(x: torch.Tensor, gated: bool) -> torch.Tensor:
    """gated: x [rows, 2F] = [a | b] -> bf16(gelu_new(a)) * b [rows, F];  else quick_gelu(x)."""
    _req(x, torch.bfloat16, "x")
    x = x.contiguous()
    F_ = x.shape[1] // 2 if gated else x.shape[1]
    out = torch.empty((x.shape[0], F_), dtype=torch.bfloat16, device=x.device)
    call("fluxmi_act_mul", _p(x), _p(out), x.shape[0], F_, x.stride(0), out.stride(0), 0 if gated else 1, _stream())
    return out


def text_attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, L: int, heads: int, scale: float = 1.0, causal: bool = False,
                   rel_bias: Optional[torch.Tensor] = None, v_bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k [Lp, >= heads*64] (views with the same row stride), vt [heads*64, Lp] -> out [Lp, heads*64]; see include/fluxmi.h."""
    _req(q, torch.bfloat16, "q")
    _req(k, torch.bfloat16, "k")
    _req(vt, torch.bfloat16, "vt")
    Lp = q.shape[0]
    if q.stride(0) != k.stride(0) or q.stride(1) != 1 or k.stride(1) != 1 or vt.stride(1) != 1:
        raise ValueError("text_attention: q and k must share the row stride; innermost strides must be 1")
    if out is None:
        out = torch.zeros((Lp, heads * 64), dtype=torch.bfloat16, device=q.device)
    if rel_bias is not None:
        _req(rel_bias, torch.float32, "rel_bias")
        rel_bias = rel_bias.contiguous()
    call("fluxmi_text_attention", _p(q), _p(k), q.stride(0), _p(vt), vt.stride(0), _p(out), out.stride(0),
         _p(rel_bias) if rel_bias is not None else None, rel_bias.shape[1] if rel_bias is not None else 0,
         _p(v_bias) if v_bias is not None else None, float(scale), int(causal), int(L), Lp, heads, _stream())
    return out
