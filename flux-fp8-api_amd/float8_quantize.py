"""F8Linear on MI355X: the reference's fp8 linear operator surface (float8_quantize.py of
aredden/flux-fp8-api) backed by libfluxmi's hand-written gfx950 kernels.

Same names, constructor arguments, buffers (state-dict format!) and state machine as the reference:
  F8Linear(in_features, out_features, bias, device, dtype, float8_dtype, float_weight, float_bias,
           num_scale_trials, input_float8_dtype)                       reference float8_quantize.py:30-89
  .forward / .from_linear / .quantize_weight / .set_weight_tensor / .quantize_input /.amax_to_scale
  buffers float8_data, scale, scale_reciprocal, input_scale, input_scale_reciprocal
  attrs  input_scale_initialized, weight_initialized, trial_index, input_amax_trials
  recursive_swap_linears, swap_to_cublaslinear, quantize_flow_transformer_and_dispatch_float8

Differences by design (MI355X-first):
  * every arithmetic step runs in libfluxmi (torch._scaled_mm is never called); no CPU path exists
  * scale / amax state lives in fixed device buffers that the kernels update in place (no host sync,
    stable pointers for hipGraph capture) instead of being re-bound to fresh tensors on every call
  * the import gate is "ROCm build of torch + libfluxmi present" instead of "CUDA >= 12.4"
"""
from __future__ import annotations

import torch
import torch.nn as nn

from fluxmi import ops  # raises ImportError loudly when libfluxmi.so has not been built

CublasLinear = type(None)  # fp16-only CUDA extension of the reference (float8_quantize.py:24-27): never present here


def _modulation_cls():
    from modules.flux_model import Modulation

    return Modulation


class F8Linear(nn.Module):
    def __init__(
        self,
        in_features: int,
        out_features: int,
        bias: bool = True,
        device=None,
        dtype=torch.float16,
        float8_dtype=torch.float8_e4m3fn,
        float_weight: torch.Tensor = None,
        float_bias: torch.Tensor = None,
        num_scale_trials: int = 12,
        input_float8_dtype=torch.float8_e5m2,
    ) -> None:
        super().__init__()
        if float8_dtype != torch.float8_e4m3fn:
            # the reference's default (float8_quantize.py:39); libfluxmi's MX-MFMA kernels fix the weight operand to e4m3fn
            raise ValueError(f"fluxmi F8Linear: weights must be torch.float8_e4m3fn, got {float8_dtype}")
        self.in_features = in_features
        self.out_features = out_features
        self.float8_dtype = float8_dtype
        self.input_float8_dtype = input_float8_dtype
        self.input_scale_initialized = False
        self.weight_initialized = False
        self.max_value = torch.finfo(self.float8_dtype).max
        self.input_max_value = torch.finfo(self.input_float8_dtype).max
        factory_kwargs = {"dtype": dtype, "device": device}
        if float_weight is None:
            self.weight = nn.Parameter(torch.empty((out_features, in_features), **factory_kwargs), requires_grad=False)
        else:
            self.weight = nn.Parameter(float_weight, requires_grad=False)
        if float_bias is None:
            if bias:
                self.bias = nn.Parameter(torch.empty(out_features, **factory_kwargs), requires_grad=False)
            else:
                self.register_parameter("bias", None)
        else:
            self.bias = nn.Parameter(float_bias, requires_grad=False)
        self.num_scale_trials = num_scale_trials
        self.input_amax_trials = torch.zeros(num_scale_trials, requires_grad=False, device=device, dtype=torch.float32)
        self.trial_index = 0
        self.register_buffer("scale", None)
        self.register_buffer("input_scale", None)
        self.register_buffer("float8_data", None)
        self.register_buffer("scale_reciprocal", None)
        self.register_buffer("input_scale_reciprocal", None)
        self._amax_tmp = None

    # ---- device state ------------------------------------------------------------------------
    def _ensure_state(self, device):
        """Fixed-address fp32 scalars the kernels write into (0-dim like the reference's buffers)."""
        device = torch.device(device)
        for name in ("scale", "scale_reciprocal", "input_scale", "input_scale_reciprocal"):
            t = getattr(self, name)
            if t is None or t.device != device:
                new = torch.ones((), dtype=torch.float32, device=device)
                if t is not None:
                    new.copy_(t.float())
                setattr(self, name, new)
        if self.input_amax_trials.device != device:
            self.input_amax_trials = self.input_amax_trials.to(device)
        if self._amax_tmp is None or self._amax_tmp.device != device:
            self._amax_tmp = torch.zeros((), dtype=torch.float32, device=device)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self.input_amax_trials.is_meta:
            self.input_amax_trials = torch.zeros(self.num_scale_trials, dtype=torch.float32)
        self.input_amax_trials = fn(self.input_amax_trials)
        if self._amax_tmp is not None:
            self._amax_tmp = fn(self._amax_tmp)
        return out

    # ---- reference API --------------------------------------------------------------------------
    def amax_to_scale(self, amax, max_val):
        return (max_val / torch.clamp(amax, min=1e-12)).clamp(max=max_val)

    def to_fp8_saturated(self, x, scale, max_val):
        """Kept for API parity; returns the fp8 tensor (the kernel fuses product, clamp and cast)."""
        return ops.quantize_act(x, scale, ops.fmt_of(self.input_float8_dtype if max_val == self.input_max_value else self.float8_dtype))

    def quantize_weight(self):
        """reference float8_quantize.py:195-207, on device."""
        if self.weight_initialized:
            return
        w = self.weight.data
        if not w.is_cuda:
            raise RuntimeError("F8Linear.quantize_weight: weight must be on the GPU (libfluxmi has no CPU path)")
        self._ensure_state(w.device)
        q, s, r = ops.quantize_weight(w.to(torch.bfloat16) if w.dtype != torch.bfloat16 else w, ops.fmt_of(self.float8_dtype))
        self.float8_data = q
        self.scale.copy_(s)
        self.scale_reciprocal.copy_(r)
        self.weight.data = torch.zeros(1, dtype=w.dtype, device=w.device, requires_grad=False)
        self.weight_initialized = True

    def set_weight_tensor(self, tensor: torch.Tensor):
        self.weight.data = tensor
        self.weight_initialized = False
        self.quantize_weight()

    def quantize_input(self, x: torch.Tensor):
        """The 12-trial running-amax calibration (reference float8_quantize.py:220-246); scales stay on device."""
        fmt = ops.fmt_of(self.input_float8_dtype)
        self._ensure_state(x.device)
        if not self.input_scale_initialized:
            if self.trial_index < self.num_scale_trials:
                self._amax_tmp.zero_()
                ops.amax(x, self._amax_tmp)
                ops.calib_update(self._amax_tmp, self.input_amax_trials, self.input_scale, self.input_scale_reciprocal,
                                 self.trial_index, self.num_scale_trials, self.input_max_value)
                self.trial_index += 1
            else:
                ops.calib_update(self._amax_tmp, self.input_amax_trials, self.input_scale, self.input_scale_reciprocal,
                                 self.num_scale_trials, self.num_scale_trials, self.input_max_value)
                self.input_scale_initialized = True
        return ops.quantize_act(x, self.input_scale, fmt)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """reference float8_quantize.py:272-296."""
        if not x.is_cuda:
            raise RuntimeError("F8Linear.forward: input must be on the GPU (libfluxmi has no CPU path)")
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        x8 = self.quantize_input(x)
        prev_dims = x8.shape[:-1]
        x8 = x8.reshape(-1, self.in_features)
        out = ops.linear(x8, self.float8_data, self.bias, self.input_scale_reciprocal, self.scale_reciprocal)
        out = out.view(*prev_dims, self.out_features)
        return out if self.weight.dtype == torch.bfloat16 else out.to(self.weight.dtype)

    def reset_parameters(self) -> None:
        """Re-draw the weight and quantise it again (same role as the reference's float8_quantize.py:248-271).  Not on the hot
        path: nn.Linear's default init is taken from a throw-away nn.Linear on the weight's device, then the calibration state
        starts over because the old input scales belong to the old weight."""
        w = self.weight
        dev, dt = (w.device, w.dtype)
        fresh = nn.Linear(self.in_features, self.out_features, bias=self.bias is not None, device=dev, dtype=torch.float32)
        self.weight = nn.Parameter(fresh.weight.detach().to(dt), requires_grad=False)
        if self.bias is not None:
            self.bias = nn.Parameter(fresh.bias.detach().to(self.bias.dtype), requires_grad=False)
        self.weight_initialized = False
        self.input_scale_initialized = False
        self.trial_index = 0
        self.input_amax_trials.zero_()
        self.quantize_weight()

    # ---- prequantised checkpoints (state-dict format of reference float8_quantize.py:91-193) -----
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        if "weight" not in sd:
            raise RuntimeError("Weight tensor not found or has incorrect shape in state dict")
        full_shape = (self.out_features, self.in_features)
        if sd.get("float8_data") is None:
            if tuple(sd["weight"].shape) != full_shape:
                raise RuntimeError(f"Weight tensor not found or has incorrect shape in state dict: {sd.keys()}")
            self._parameters["weight"] = nn.Parameter(sd["weight"], requires_grad=False)
            if "bias" in sd:
                self._parameters["bias"] = nn.Parameter(sd["bias"], requires_grad=False)
            self.weight_initialized = False
            if sd["weight"].is_cuda:
                self.quantize_weight()
            return
        if tuple(sd["float8_data"].shape) != full_shape or bool((sd["weight"] != 0).any()):
            raise RuntimeError(f"Weight tensor not found or has incorrect shape in state dict: {sd.keys()}")
        w = sd["weight"]
        self._buffers["float8_data"] = sd["float8_data"]
        self._parameters["weight"] = nn.Parameter(torch.zeros(1, dtype=w.dtype, device=w.device), requires_grad=False)
        if "bias" in sd:
            self._parameters["bias"] = nn.Parameter(sd["bias"], requires_grad=False)
        self.weight_initialized = True
        dev = sd["float8_data"].device
        have_w = "scale" in sd and "scale_reciprocal" in sd
        have_in = "input_scale" in sd and "input_scale_reciprocal" in sd
        if have_w:
            self.scale = sd["scale"].float().reshape(()).to(dev).clone()
            self.scale_reciprocal = sd["scale_reciprocal"].float().reshape(()).to(dev).clone()
        if have_w and have_in:
            self.input_scale = sd["input_scale"].float().reshape(()).to(dev).clone()
            self.input_scale_reciprocal = sd["input_scale_reciprocal"].float().reshape(()).to(dev).clone()
            self.input_scale_initialized = True
            self.trial_index = self.num_scale_trials
            if self.input_amax_trials.is_meta:  # module built under torch.device("meta") (util.load_flow_model): not part of the state dict
                self.input_amax_trials = torch.zeros(self.num_scale_trials, dtype=torch.float32, device=dev)
        else:
            self.input_scale_initialized = False
            self.trial_index = 0
            self.input_amax_trials = torch.zeros(self.num_scale_trials, dtype=torch.float32, device=dev)

    @classmethod
    def from_linear(cls, linear: nn.Linear, float8_dtype=torch.float8_e4m3fn, input_float8_dtype=torch.float8_e5m2) -> "F8Linear":
        f8_lin = cls(
            in_features=linear.in_features,
            out_features=linear.out_features,
            bias=linear.bias is not None,
            device=linear.weight.device,
            dtype=linear.weight.dtype,
            float8_dtype=float8_dtype,
            float_weight=linear.weight.data,
            float_bias=(linear.bias.data if linear.bias is not None else None),
            input_float8_dtype=input_float8_dtype,
        )
        f8_lin.quantize_weight()
        return f8_lin


@torch.inference_mode()
def recursive_swap_linears(model: nn.Module, float8_dtype=torch.float8_e4m3fn, input_float8_dtype=torch.float8_e5m2,
                           quantize_modulation: bool = True, ignore_keys: list[str] = []) -> None:
    """Replace every nn.Linear below `model` by an F8Linear (reference float8_quantize.py:320-369)."""
    Modulation = _modulation_cls()
    for name, child in model.named_children():
        if name in ignore_keys:
            continue
        if isinstance(child, Modulation) and not quantize_modulation:
            continue
        if isinstance(child, nn.Linear) and not isinstance(child, F8Linear):
            setattr(model, name, F8Linear.from_linear(child, float8_dtype=float8_dtype, input_float8_dtype=input_float8_dtype))
        else:
            recursive_swap_linears(child, float8_dtype=float8_dtype, input_float8_dtype=input_float8_dtype,
                                   quantize_modulation=quantize_modulation, ignore_keys=ignore_keys)


@torch.inference_mode()
def swap_to_cublaslinear(model: nn.Module):
    """fp16-flow-only path of the reference (float8_quantize.py:372-392); the bf16 flow never takes it."""
    return


@torch.inference_mode()
def quantize_flow_transformer_and_dispatch_float8(
    flow_model: nn.Module,
    device=torch.device("cuda"),
    float8_dtype=torch.float8_e4m3fn,
    input_float8_dtype=torch.float8_e5m2,
    offload_flow=False,
    swap_linears_with_cublaslinear=True,
    flow_dtype=torch.float16,
    quantize_modulation: bool = True,
    quantize_flow_embedder_layers: bool = True,
) -> nn.Module:
    """Block-by-block move-to-device + quantise (reference float8_quantize.py:395-496).

    Which layers become F8Linear is identical to the reference: every Linear of every double/single
    block (Modulation only if quantize_modulation), the embedders img_in/txt_in/time_in/vector_in/
    guidance_in only if quantize_flow_embedder_layers, final_layer never."""
    kw = dict(float8_dtype=float8_dtype, input_float8_dtype=input_float8_dtype, quantize_modulation=quantize_modulation)
    for module in list(flow_model.double_blocks) + list(flow_model.single_blocks):
        module.to(device)
        module.eval()
        recursive_swap_linears(module, **kw)
    for name in ("vector_in", "img_in", "txt_in", "time_in", "guidance_in", "final_layer", "pe_embedder"):
        m_extra = getattr(flow_model, name)
        if m_extra is None:
            continue
        m_extra.to(device)
        m_extra.eval()
        if isinstance(m_extra, nn.Linear) and not isinstance(m_extra, F8Linear):
            if quantize_flow_embedder_layers:
                setattr(flow_model, name, F8Linear.from_linear(m_extra, float8_dtype=float8_dtype, input_float8_dtype=input_float8_dtype))
        elif name != "final_layer" and quantize_flow_embedder_layers:
            recursive_swap_linears(m_extra, **kw)
    if hasattr(flow_model, "_invalidate_engine"):
        flow_model._invalidate_engine()
    # offload_flow is meaningless with 288 GB of HBM: accepted and ignored
    return flow_model
