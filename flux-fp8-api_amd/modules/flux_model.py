"""Flux flow transformer on MI355X: the module tree of the reference (modules/flux_model.py of
aredden/flux-fp8-api) as a thin parameter container over the native fluxmi engine.

Kept from the reference (so that float8_quantize / lora_loading / pipeline code that WALKS the tree by
name keeps working): class names, constructor arguments, attribute names
(double_blocks[i].img_mod.lin, .img_attn.qkv/.norm.query_norm.scale/.proj, .img_mlp[0|2], ...,
single_blocks[i].linear1/.linear2/.norm/.modulation.lin, final_layer.linear/.adaLN_modulation[1],
img_in, txt_in, time_in, vector_in, guidance_in, pe_embedder), the BFL state-dict key layout,
`Flux.forward(img, img_ids, txt, txt_ids, timesteps, y, guidance)` and the LoRA bookkeeping methods.

Different by design: the modules hold weights only.  `Flux.forward` hands raw device pointers to the C++
engine (csrc/engine.hip), which runs the whole step on hand-written gfx950 kernels -- fused and
hipGraph-captured once every F8Linear input scale is frozen, unfused (reference op order) while the
12-trial calibration of float8_quantize.py:220-246 is still running.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from collections import namedtuple
from typing import TYPE_CHECKING, List, Optional

import torch
from pydantic import BaseModel
from torch import Tensor, nn

from fluxmi import _lib, ops

if TYPE_CHECKING:
    from util import ModelSpec


class FluxParams(BaseModel):  # reference flux_model.py:24-36
    in_channels: int
    vec_in_dim: int
    context_in_dim: int
    hidden_size: int
    mlp_ratio: float
    num_heads: int
    depth: int
    depth_single_blocks: int
    axes_dim: list[int]
    theta: int
    qkv_bias: bool
    guidance_embed: bool


ModulationOut = namedtuple("ModulationOut", ["shift", "scale", "gate"])


def _f8():
    from float8_quantize import F8Linear

    return F8Linear


def _lin(i, o, bias=True, f8=False):
    return _f8()(in_features=i, out_features=o, bias=bias) if f8 else nn.Linear(i, o, bias=bias)


def timestep_embedding(t: Tensor, dim, max_period=10000, time_factor: float = 1000.0):
    """reference flux_model.py:95-116 (device kernel; `t` is rounded through its own dtype like the reference)."""
    freqs = ops.timestep_freqs_host(dim // 2, max_period).to(t.device)
    return ops.timestep_embedding(t.to(torch.bfloat16), freqs, time_factor)


class EmbedND(nn.Module):  # reference flux_model.py:68-92
    def __init__(self, dim: int, theta: int, axes_dim: list[int], dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.dim, self.theta, self.axes_dim, self.dtype = dim, theta, axes_dim, dtype

    def forward(self, ids: Tensor) -> Tensor:
        """Returns the reference layout [B,1,L,dim/2,2,2] = [[cos,-sin],[sin,cos]] built from the device table."""
        pe = ops.rope_table(ids.to(torch.bfloat16), self.axes_dim, self.theta)  # [B,L,P,2]
        c, s = pe[..., 0], pe[..., 1]
        return torch.stack((c, -s, s, c), dim=-1).reshape(*c.shape, 2, 2).unsqueeze(1).to(self.dtype)


class MLPEmbedder(nn.Module):  # reference flux_model.py:119-155
    def __init__(self, in_dim: int, hidden_dim: int, prequantized: bool = False, quantized=False):
        super().__init__()
        self.in_layer = _lin(in_dim, hidden_dim, f8=prequantized and quantized)
        self.silu = nn.SiLU()
        self.out_layer = _lin(hidden_dim, hidden_dim, f8=prequantized and quantized)


class RMSNorm(nn.Module):  # reference flux_model.py:158-164
    def __init__(self, dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim), requires_grad=False)


class QKNorm(nn.Module):  # reference flux_model.py:167-176
    def __init__(self, dim: int):
        super().__init__()
        self.query_norm = RMSNorm(dim)
        self.key_norm = RMSNorm(dim)


class SelfAttention(nn.Module):  # reference flux_model.py:179-227
    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, prequantized: bool = False):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = _lin(dim, dim * 3, bias=qkv_bias, f8=prequantized)
        self.norm = QKNorm(dim // num_heads)
        self.proj = _lin(dim, dim, f8=prequantized)


class Modulation(nn.Module):  # reference flux_model.py:233-257
    def __init__(self, dim: int, double: bool, quantized_modulation: bool = False):
        super().__init__()
        self.is_double = double
        self.multiplier = 6 if double else 3
        self.lin = _lin(dim, self.multiplier * dim, f8=quantized_modulation)
        self.act = nn.SiLU()


class DoubleStreamBlock(nn.Module):  # reference flux_model.py:260-400
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float, qkv_bias: bool = False,
                 dtype: torch.dtype = torch.float16, quantized_modulation: bool = False, prequantized: bool = False):
        super().__init__()
        self.dtype, self.num_heads, self.hidden_size = dtype, num_heads, hidden_size
        mlp_hidden_dim = int(hidden_size * mlp_ratio)
        for s in ("img", "txt"):
            setattr(self, f"{s}_mod", Modulation(hidden_size, double=True, quantized_modulation=quantized_modulation))
            setattr(self, f"{s}_norm1", nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6))
            setattr(self, f"{s}_attn", SelfAttention(dim=hidden_size, num_heads=num_heads, qkv_bias=qkv_bias, prequantized=prequantized))
            setattr(self, f"{s}_norm2", nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6))
            setattr(self, f"{s}_mlp", nn.Sequential(_lin(hidden_size, mlp_hidden_dim, f8=prequantized), nn.GELU(approximate="tanh"),
                                                    _lin(mlp_hidden_dim, hidden_size, f8=prequantized)))


class SingleStreamBlock(nn.Module):  # reference flux_model.py:403-485
    def __init__(self, hidden_size: int, num_heads: int, mlp_ratio: float = 4.0, qk_scale: float | None = None,
                 dtype: torch.dtype = torch.float16, quantized_modulation: bool = False, prequantized: bool = False):
        super().__init__()
        self.dtype, self.hidden_dim, self.hidden_size, self.num_heads = dtype, hidden_size, hidden_size, num_heads
        self.mlp_hidden_dim = int(hidden_size * mlp_ratio)
        self.linear1 = _lin(hidden_size, hidden_size * 3 + self.mlp_hidden_dim, f8=prequantized)
        self.linear2 = _lin(hidden_size + self.mlp_hidden_dim, hidden_size, f8=prequantized)
        self.norm = QKNorm(hidden_size // num_heads)
        self.pre_norm = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp_act = nn.GELU(approximate="tanh")
        self.modulation = Modulation(hidden_size, double=False, quantized_modulation=quantized_modulation and prequantized)


class LastLayer(nn.Module):  # reference flux_model.py:488-503
    def __init__(self, hidden_size: int, patch_size: int, out_channels: int):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class Flux(nn.Module):
    """Transformer model for flow matching on sequences (reference flux_model.py:506-734)."""

    MAX_ENGINE_BATCH = 32  # samples per engine pass (csrc/engine.hip FLUXMI_ENGINE_MAX_BATCH)

    def __init__(self, config: "ModelSpec", dtype: torch.dtype = torch.float16):
        super().__init__()
        self.dtype = dtype
        self.params = p = config.params
        self.in_channels = self.out_channels = p.in_channels
        self.loras: List = []
        preq = config.prequantized_flow
        q_emb = config.quantize_flow_embedder_layers and preq
        q_mod = config.quantize_modulation and preq
        if p.hidden_size % p.num_heads != 0:
            raise ValueError(f"Hidden size {p.hidden_size} must be divisible by num_heads {p.num_heads}")
        pe_dim = p.hidden_size // p.num_heads
        if sum(p.axes_dim) != pe_dim:
            raise ValueError(f"Got {p.axes_dim} but expected positional dim {pe_dim}")
        self.hidden_size, self.num_heads = p.hidden_size, p.num_heads
        self.pe_embedder = EmbedND(dim=pe_dim, theta=p.theta, axes_dim=p.axes_dim, dtype=self.dtype)
        self.img_in = _lin(self.in_channels, self.hidden_size, f8=q_emb)
        self.time_in = MLPEmbedder(256, self.hidden_size, prequantized=preq, quantized=q_emb)
        self.vector_in = MLPEmbedder(p.vec_in_dim, self.hidden_size, prequantized=preq, quantized=q_emb)
        self.guidance_in = MLPEmbedder(256, self.hidden_size, prequantized=preq, quantized=q_emb) if p.guidance_embed else nn.Identity()
        self.txt_in = _lin(p.context_in_dim, self.hidden_size, f8=q_emb)
        self.double_blocks = nn.ModuleList([
            DoubleStreamBlock(self.hidden_size, self.num_heads, mlp_ratio=p.mlp_ratio, qkv_bias=p.qkv_bias, dtype=self.dtype,
                              quantized_modulation=q_mod, prequantized=preq) for _ in range(p.depth)])
        self.single_blocks = nn.ModuleList([
            SingleStreamBlock(self.hidden_size, self.num_heads, mlp_ratio=p.mlp_ratio, dtype=self.dtype,
                              quantized_modulation=q_mod, prequantized=preq) for _ in range(p.depth_single_blocks)])
        self.final_layer = LastLayer(self.hidden_size, 1, self.out_channels)
        self.requires_grad_(False)
        self._engine = None
        self._engine_keep = None
        self._engine_device = None
        self._lock = threading.Lock()  # the C handle is not re-entrant (the reference's api.py calls from a threadpool)
        self._prep_key = None
        self._amax_xchg = None  # (device array, ctypes callback) of the batch-sharded calibration exchange, see enable_amax_exchange

    # ---- engine plumbing ---------------------------------------------------------------------------------
    def linear_modules(self) -> List[nn.Module]:
        """Linear layers in the canonical order of include/fluxmi.h."""
        mods = [self.img_in, self.time_in.in_layer, self.time_in.out_layer, self.vector_in.in_layer, self.vector_in.out_layer]
        if self.params.guidance_embed:
            mods += [self.guidance_in.in_layer, self.guidance_in.out_layer]
        mods.append(self.txt_in)
        for b in self.double_blocks:
            for s in ("img", "txt"):
                a, m = getattr(b, f"{s}_attn"), getattr(b, f"{s}_mlp")
                mods += [getattr(b, f"{s}_mod").lin, a.qkv, a.proj, m[0], m[2]]
        for b in self.single_blocks:
            mods += [b.modulation.lin, b.linear1, b.linear2]
        mods += [self.final_layer.adaLN_modulation[1], self.final_layer.linear]
        return mods

    def f8_modules(self):
        F8 = _f8()
        return [m for m in self.linear_modules() if isinstance(m, F8)]

    def _invalidate_engine(self):
        with self._lock:
            if self._engine is not None:
                _lib.call("fluxmi_engine_destroy", self._engine)
            self._engine, self._engine_keep, self._prep_key, self._engine_device = None, None, None, None

    def __del__(self):
        try:
            if getattr(self, "_engine", None) is not None:
                _lib.lib.fluxmi_engine_destroy(self._engine)
        except Exception:
            pass

    def _linear_table(self, device):
        F8 = _f8()
        mods = self.linear_modules()
        arr = (_lib.Linear * len(mods))()
        keep = []
        for i, m in enumerate(mods):
            L = arr[i]
            if isinstance(m, F8):
                if not m.weight_initialized:
                    m.to(device)
                    m.quantize_weight()
                m._ensure_state(device)
                if m.float8_data.device != device:
                    m.to(device)
                L.weight, L.kind = m.float8_data.data_ptr(), 1
                L.w_scale_recip, L.in_scale = m.scale_reciprocal.data_ptr(), m.input_scale.data_ptr()
                L.in_scale_recip, L.amax_trials = m.input_scale_reciprocal.data_ptr(), m.input_amax_trials.data_ptr()
                L.in_fmt = ops.fmt_of(m.input_float8_dtype)
                L.N, L.K = m.out_features, m.in_features
                keep += [m.float8_data, m.scale_reciprocal, m.input_scale, m.input_scale_reciprocal, m.input_amax_trials]
            else:
                if m.weight.device != device or m.weight.dtype != torch.bfloat16:
                    m.to(device=device, dtype=torch.bfloat16)
                w = m.weight.data.contiguous()
                L.weight, L.kind, L.in_fmt = w.data_ptr(), 0, _lib.E5M2
                L.N, L.K = w.shape
                keep.append(w)
            b = m.bias
            if b is not None:
                bb = b.data.to(device=device, dtype=torch.bfloat16).contiguous()
                L.bias = bb.data_ptr()
                keep.append(bb)
        return arr, keep

    def _norm_table(self, device):
        ts = []
        for b in self.double_blocks:
            ts += [b.img_attn.norm.query_norm.scale, b.img_attn.norm.key_norm.scale, b.txt_attn.norm.query_norm.scale, b.txt_attn.norm.key_norm.scale]
        for b in self.single_blocks:
            ts += [b.norm.query_norm.scale, b.norm.key_norm.scale]
        keep = [t.data.to(device=device, dtype=torch.bfloat16).contiguous() for t in ts]
        arr = (C.c_void_p * len(keep))(*[k.data_ptr() for k in keep])
        return arr, keep

    def _ensure_engine(self, device):
        if self._engine is not None:
            return
        if device.type != "cuda":
            raise RuntimeError("Flux (fluxmi): the model must run on a GPU; there is no CPU path")
        p = self.params
        if p.hidden_size // p.num_heads != 128:
            raise ValueError("fluxmi supports head_dim 128 (every FLUX.1 variant); got %d" % (p.hidden_size // p.num_heads))
        d = _lib.ModelDesc()
        d.hidden, d.heads, d.mlp_hidden = p.hidden_size, p.num_heads, int(p.hidden_size * p.mlp_ratio)
        d.depth, d.depth_single, d.in_channels = p.depth, p.depth_single_blocks, p.in_channels
        d.vec_in, d.ctx_in, d.guidance_embed = p.vec_in_dim, p.context_in_dim, int(p.guidance_embed)
        d.axes_dim = (C.c_int * 3)(*p.axes_dim)
        d.theta = p.theta
        f8 = self.f8_modules()
        d.num_trials = f8[0].num_scale_trials if f8 else 12
        lin, keep_l = self._linear_table(device)
        nrm, keep_n = self._norm_table(device)
        h = C.c_void_p()
        _lib.call("fluxmi_engine_create", C.byref(d), lin, len(lin), nrm, len(nrm), C.byref(h))
        om, ax = ops.rope_tables_host(p.axes_dim, p.theta)
        fr = ops.timestep_freqs_host(128)
        _lib.call("fluxmi_engine_set_tables", h, fr.numpy().ctypes.data_as(C.POINTER(C.c_float)),
                  om.numpy().ctypes.data_as(C.POINTER(C.c_float)), ax.numpy().ctypes.data_as(C.POINTER(C.c_int)))
        self._engine, self._engine_keep, self._prep_key, self._engine_device = h, (keep_l, keep_n, lin, nrm), None, device
        if self._amax_xchg is not None:
            self._install_amax_exchange()

    def rebind_weights(self):
        """Call after weight surgery (LoRA fuse, set_weight_tensor) so the engine sees the new pointers."""
        if self._engine is None:
            return
        dev = self._engine_device  # the device the engine was created on (not necessarily torch's current device)
        lin, keep_l = self._linear_table(dev)
        with self._lock:
            _lib.call("fluxmi_engine_rebind", self._engine, lin, len(lin))
            self._engine_keep = (keep_l, self._engine_keep[1], lin, self._engine_keep[3])

    def _prepare(self, img, img_ids, txt_ids, txt):
        B, Li, _ = img.shape
        Lt = txt.shape[1]
        # cheap (two small copies + one table kernel; the workspace is only re-allocated when the shape changes)
        ii = img_ids.to(torch.bfloat16).contiguous()
        ti = txt_ids.to(torch.bfloat16).contiguous()
        _lib.call("fluxmi_engine_prepare", self._engine, B, Li, Lt, ops._p(ii), ops._p(ti), ops._stream())

    # ---- batch-sharded calibration ------------------------------------------------------------------
    def enable_amax_exchange(self, reduce_fn=None):
        """Keep the F8Linear input scales of batch-sharded replicas IDENTICAL to those of the whole batch on one GPU: the reference
        takes amax over the whole batch (float8_quantize.py:227), so during the calibrating steps every layer's running amax is
        MAX-reduced across the ranks before its scale update (fluxmi_engine_set_amax_exchange).  `reduce_fn(tensor)` performs the
        in-place reduction of a small fp32 device tensor on the current stream; default: torch.distributed all_reduce(MAX) over
        RCCL.  Pass reduce_fn=False to uninstall."""
        if reduce_fn is False:
            self._amax_xchg = None
            if self._engine is not None:
                with self._lock:
                    _lib.call("fluxmi_engine_set_amax_exchange", self._engine, None, 0, None, None)
            return
        if reduce_fn is None:
            import torch.distributed as td

            reduce_fn = lambda t: td.all_reduce(t, op=td.ReduceOp.MAX)
        self._amax_xchg = {"fn": reduce_fn, "buf": None, "cb": None, "error": None}
        if self._engine is not None:
            with self._lock:
                self._install_amax_exchange()

    def _install_amax_exchange(self):
        x = self._amax_xchg
        n = len(self.linear_modules())
        x["buf"] = torch.zeros(n, dtype=torch.float32, device=self._engine_device)

        def hook(user, first, count, stream):
            try:  # called by the engine between the amax reduction of layers [first, first+count) and their scale update
                x["fn"](x["buf"][first:first + count])
                return 0
            except Exception as e:  # an exception must not cross the C ABI
                x["error"] = e
                return 1

        x["cb"] = _lib.AMAX_HOOK(hook)
        _lib.call("fluxmi_engine_set_amax_exchange", self._engine, ops._p(x["buf"]), n, x["cb"], None)

    # ---- calibration bookkeeping (mirrors F8Linear.trial_index / input_scale_initialized) -----------------
    def calibration_state(self):
        f8 = self.f8_modules()
        if not f8:
            return None, 0
        return all(m.input_scale_initialized for m in f8), min(m.trial_index for m in f8)

    def _advance_calibration(self, new_trial_index: int):
        for m in self.f8_modules():
            if new_trial_index > m.num_scale_trials:
                m.trial_index, m.input_scale_initialized = m.num_scale_trials, True
            else:
                m.trial_index = new_trial_index

    def _trial_counter(self):
        frozen, t = self.calibration_state()
        if frozen is None:
            return None
        f8 = self.f8_modules()
        return f8[0].num_scale_trials + 1 if frozen else t

    # ---- LoRA bookkeeping (reference flux_model.py:621-670) --------------------------------------------------
    def get_lora(self, identifier: str):
        for lora in self.loras:
            if lora.path == identifier or lora.name == identifier:
                return lora

    def has_lora(self, identifier: str):
        return self.get_lora(identifier) is not None

    def load_lora(self, path: str, scale: float, name: str = None):
        from lora_loading import LoraWeights, apply_lora_to_model, remove_lora_from_module

        if self.has_lora(path):
            lora = self.get_lora(path)
            if lora.scale != scale:
                remove_lora_from_module(self, lora, lora.scale)
                apply_lora_to_model(self, lora, scale)
                lora.scale = scale
        else:
            _, lora = apply_lora_to_model(self, path, scale, return_lora_resolved=True)
            self.loras.append(LoraWeights(lora, path if isinstance(path, str) else (name or "lora"), name, scale))
        self.rebind_weights()

    def unload_lora(self, path_or_identifier: str):
        from lora_loading import remove_lora_from_module

        for idx, lora_ in enumerate(list(self.loras)):
            if lora_.path == path_or_identifier or lora_.name == path_or_identifier:
                remove_lora_from_module(self, lora_.weights, lora_.scale)
                self.loras.pop(idx)
                self.rebind_weights()
                return True
        return False

    # ---- forward / denoise ---------------------------------------------------------------------------------------
    @torch.inference_mode()
    def forward(self, img: Tensor, img_ids: Tensor, txt: Tensor, txt_ids: Tensor, timesteps: Tensor, y: Tensor,
                guidance: Tensor | None = None, mode: Optional[int] = None) -> Tensor:
        """One denoise-step evaluation (reference flux_model.py:672-716).  mode=None picks what the reference would do:
        calibrating (unfused) while any F8Linear still has trials to record, fused once frozen."""
        if img.ndim != 3 or txt.ndim != 3:
            raise ValueError("Input img and txt tensors must have 3 dimensions.")
        if self.params.guidance_embed and guidance is None:
            raise ValueError("Didn't get guidance strength for guidance distilled model.")
        bf = lambda t: t.to(torch.bfloat16).contiguous()
        img, txt, y, timesteps = bf(img), bf(txt), bf(y), bf(timesteps)
        guidance = bf(guidance) if guidance is not None else None
        self._ensure_engine(img.device)
        with self._lock:
            self._prepare(img, img_ids, txt_ids, txt)
            trial = self._trial_counter()
            if mode is None:
                if trial is None:
                    mode = 2
                elif trial <= self.f8_modules()[0].num_scale_trials:
                    mode = 0
                else:
                    mode = 1 if self._all_block_linears_f8() else 2
            pred = torch.empty(img.shape[0], img.shape[1], self.out_channels, dtype=torch.bfloat16, device=img.device)
            _lib.call("fluxmi_engine_forward", self._engine, ops._p(img), ops._p(txt), ops._p(y), ops._p(timesteps), ops._p(guidance),
                      ops._p(pred), mode, trial if mode == 0 else 0, ops._stream())
            if mode == 0:
                self._advance_calibration(trial + 1)
        return pred if self.dtype == torch.bfloat16 else pred.to(self.dtype)

    def _all_block_linears_f8(self):
        F8 = _f8()
        for b in self.double_blocks:
            for s in ("img", "txt"):
                a, m = getattr(b, f"{s}_attn"), getattr(b, f"{s}_mlp")
                if not all(isinstance(x, F8) for x in (a.qkv, a.proj, m[0], m[2])):
                    return False
        return all(isinstance(b.linear1, F8) and isinstance(b.linear2, F8) for b in self.single_blocks)

    @torch.inference_mode()
    def denoise(self, img: Tensor, img_ids: Tensor, txt: Tensor, txt_ids: Tensor, y: Tensor, timesteps: List[float],
                guidance: float = 3.5, use_graph: bool = True) -> Tensor:
        """The Euler loop of FluxPipeline.generate (reference flux_pipeline.py:619-651) run natively: calibrating
        steps unfused, every later step one replay of a captured hipGraph.  Returns the final latent tokens."""
        bf = lambda t: t.to(torch.bfloat16).contiguous()
        if img.shape[0] > self.MAX_ENGINE_BATCH:
            # the engine takes at most 32 samples per pass (workspace / modulation-table size); the reference has no num_images limit, so
            # larger batches run as consecutive passes (samples never interact).  EQUAL passes: the engine re-allocates its workspace and
            # re-captures its graph whenever the batch size changes, so 40 = 20 + 20, not 32 + 8.  Only frozen models: a calibrating pass
            # per chunk would advance the F8Linear trial counters once per chunk instead of once per step.
            if self.calibration_state()[0] is False:
                raise ValueError(f"fluxmi: batches larger than {self.MAX_ENGINE_BATCH} need frozen F8Linear input scales (run the calibration warm-up first)")
            B = img.shape[0]
            n_pass = -(-B // self.MAX_ENGINE_BATCH)
            per = -(-B // n_pass)
            outs = []
            for i in range(0, B, per):
                sl = slice(i, min(i + per, B))
                pad = per - (sl.stop - sl.start)  # a short last pass is padded with copies of its last sample (same B -> same graph)
                pick = lambda t: torch.cat([t[sl], t[sl.stop - 1:sl.stop].expand(pad, *t.shape[1:])], 0) if pad else t[sl]
                o = self.denoise(pick(img), pick(img_ids), pick(txt), pick(txt_ids), pick(y), timesteps, guidance=guidance, use_graph=use_graph)
                outs.append(o[:per - pad])
            return torch.cat(outs, 0)
        img = bf(img).clone()
        txt, y = bf(txt), bf(y)
        self._ensure_engine(img.device)
        with self._lock:
            self._prepare(img, img_ids, txt_ids, txt)
            trial = self._trial_counter()
            t_io = C.c_int(trial if trial is not None else 0)
            ts = (C.c_double * len(timesteps))(*[float(t) for t in timesteps])
            _lib.call("fluxmi_engine_denoise", self._engine, ops._p(img), ops._p(txt), ops._p(y), float(guidance), ts,
                      len(timesteps) - 1, C.byref(t_io), int(use_graph), ops._stream())
            if trial is not None:
                self._advance_calibration(t_io.value)
        return img

    @classmethod
    def from_pretrained(cls, path: str, dtype: torch.dtype = torch.float16) -> "Flux":
        from safetensors.torch import load_file

        from util import load_config_from_path

        config = load_config_from_path(path)
        with torch.device("meta"):
            klass = cls(config=config, dtype=dtype)
            if not config.prequantized_flow:
                klass.type(dtype)
        ckpt = load_file(config.ckpt_path, device="cpu")
        klass.load_state_dict(ckpt, assign=True)
        return klass.to("cpu")
