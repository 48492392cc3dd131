"""Config surface of the reference (util.py of aredden/flux-fp8-api): ModelSpec / load_config* with the
same field names, defaults and JSON schema, so every shipped config JSON loads unchanged.

Out of the hot path (SURVEY.md §8f "next"): text encoders and VAE loading -- `load_models_from_config`
returns None for them; the pipeline then works from pre-computed embeddings and returns latents."""
from __future__ import annotations

import json
from enum import Enum
from pathlib import Path
from typing import Literal, Optional

import torch
from pydantic import BaseModel, ConfigDict

from modules.autoencoder import AutoEncoderParams
from modules.flux_model import Flux, FluxParams


class StrEnum(str, Enum):
    pass


class ModelVersion(StrEnum):
    flux_dev = "flux-dev"
    flux_schnell = "flux-schnell"


class QuantizationDtype(StrEnum):
    qfloat8 = "qfloat8"
    qint2 = "qint2"
    qint4 = "qint4"
    qint8 = "qint8"
    bfloat16 = "bfloat16"
    float16 = "float16"


class ModelSpec(BaseModel):  # reference util.py:38-79 (unknown JSON keys are ignored, as there)
    version: ModelVersion
    params: FluxParams
    ae_params: AutoEncoderParams
    ckpt_path: str | None
    clip_path: str | None = "openai/clip-vit-large-patch14"
    ae_path: str | None
    repo_id: str | None
    repo_flow: str | None
    repo_ae: str | None
    text_enc_max_length: int = 512
    text_enc_path: str | None
    text_enc_device: str | torch.device | None = "cuda:0"
    ae_device: str | torch.device | None = "cuda:0"
    flux_device: str | torch.device | None = "cuda:0"
    flow_dtype: str = "float16"
    ae_dtype: str = "bfloat16"
    text_enc_dtype: str = "bfloat16"
    num_to_quant: Optional[int] = 20
    quantize_extras: bool = False
    compile_extras: bool = False
    compile_blocks: bool = False
    flow_quantization_dtype: Optional[QuantizationDtype] = QuantizationDtype.qfloat8
    text_enc_quantization_dtype: Optional[QuantizationDtype] = QuantizationDtype.qfloat8
    ae_quantization_dtype: Optional[QuantizationDtype] = None
    clip_quantization_dtype: Optional[QuantizationDtype] = None
    offload_text_encoder: bool = False
    offload_vae: bool = False
    offload_flow: bool = False
    prequantized_flow: bool = False
    quantize_modulation: bool = True
    quantize_flow_embedder_layers: bool = False

    model_config: ConfigDict = {"arbitrary_types_allowed": True, "use_enum_values": True}


def parse_device(device) -> torch.device:
    if isinstance(device, str):
        return torch.device(device)
    if isinstance(device, torch.device):
        return device
    return torch.device("cuda:0")


def into_dtype(dtype) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    try:
        return {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[dtype]
    except KeyError:
        raise ValueError(f"Invalid dtype: {dtype}")


_warned_fp16 = False


def engine_flow_dtype(flow_dtype) -> torch.dtype:
    """The dtype the flow transformer COMPUTES in.  Every JSON the reference ships says `flow_dtype: float16` (configs/*.json:47): its
    fp16 flow is the CublasLinear + fp16-accumulate path for consumer GPUs (float8_quantize.py:372-392, the fp16 clamp of
    flux_model.py:397-399), and its README (:92) recommends bfloat16 wherever the hardware has it.  libfluxmi implements the bf16 flow
    only, so a float16 (or float32) config is accepted as is: parameters are held and every kernel runs in bf16, and tensors that cross the
    model boundary (`Flux.forward` output, the pipeline's noise / latents) are cast to the configured dtype.  Logged once."""
    global _warned_fp16
    dt = into_dtype(flow_dtype)
    if dt != torch.bfloat16 and not _warned_fp16:
        import warnings

        warnings.warn(f"fluxmi: flow_dtype={str(dt).replace('torch.', '')} requested; the MI355X engine computes the flow in bfloat16 "
                      "(inputs / outputs are cast to the requested dtype)", stacklevel=3)
        _warned_fp16 = True
    return torch.bfloat16


def into_device(device) -> torch.device:
    if isinstance(device, int):
        return torch.device(f"cuda:{device}")
    return parse_device(device)


def load_config(
    name: ModelVersion = ModelVersion.flux_dev,
    flux_path: str | None = None,
    ae_path: str | None = None,
    text_enc_path: str | None = None,
    text_enc_device=None,
    ae_device=None,
    flux_device=None,
    flow_dtype: str = "float16",
    ae_dtype: str = "bfloat16",
    text_enc_dtype: str = "bfloat16",
    num_to_quant: Optional[int] = 20,
    compile_extras: bool = False,
    compile_blocks: bool = False,
    offload_text_enc: bool = False,
    offload_ae: bool = False,
    offload_flow: bool = False,
    quant_text_enc: Optional[Literal["float8", "qint2", "qint4", "qint8"]] = None,
    quant_ae: bool = False,
    prequantized_flow: bool = False,
    quantize_modulation: bool = True,
    quantize_flow_embedder_layers: bool = False,
) -> ModelSpec:
    """Same arguments / defaults as reference util.py:122-213."""
    is_dev = name == ModelVersion.flux_dev
    return ModelSpec(
        version=name,
        repo_id="black-forest-labs/FLUX.1-dev" if is_dev else "black-forest-labs/FLUX.1-schnell",
        repo_flow="flux1-dev.sft" if is_dev else "flux1-schnell.sft",
        repo_ae="ae.sft",
        ckpt_path=flux_path,
        params=FluxParams(in_channels=64, vec_in_dim=768, context_in_dim=4096, hidden_size=3072, mlp_ratio=4.0, num_heads=24,
                          depth=19, depth_single_blocks=38, axes_dim=[16, 56, 56], theta=10_000, qkv_bias=True, guidance_embed=is_dev),
        ae_path=ae_path,
        ae_params=AutoEncoderParams(resolution=256, in_channels=3, ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                                    z_channels=16, scale_factor=0.3611, shift_factor=0.1159),
        text_enc_path=text_enc_path,
        text_enc_device=str(parse_device(text_enc_device)),
        ae_device=str(parse_device(ae_device)),
        flux_device=str(parse_device(flux_device)),
        flow_dtype=flow_dtype,
        ae_dtype=ae_dtype,
        text_enc_dtype=text_enc_dtype,
        text_enc_max_length=512 if is_dev else 256,
        num_to_quant=num_to_quant,
        compile_extras=compile_extras,
        compile_blocks=compile_blocks,
        offload_flow=offload_flow,
        offload_text_encoder=offload_text_enc,
        offload_vae=offload_ae,
        text_enc_quantization_dtype={"float8": QuantizationDtype.qfloat8, "qint2": QuantizationDtype.qint2,
                                     "qint4": QuantizationDtype.qint4, "qint8": QuantizationDtype.qint8}.get(quant_text_enc, None),
        ae_quantization_dtype=QuantizationDtype.qfloat8 if quant_ae else None,
        prequantized_flow=prequantized_flow,
        quantize_modulation=quantize_modulation,
        quantize_flow_embedder_layers=quantize_flow_embedder_layers,
    )


def load_config_from_path(path: str) -> ModelSpec:
    path_path = Path(path)
    if not path_path.exists():
        raise ValueError(f"Path {path} does not exist")
    if not path_path.is_file():
        raise ValueError(f"Path {path} is not a file")
    return ModelSpec(**json.loads(path_path.read_text()))


def load_flow_model(config: ModelSpec, state_dict=None) -> Flux:
    """reference util.py:240-256.  `state_dict` lets tests / the bench inject a synthetic BFL checkpoint."""
    dtype = into_dtype(config.flow_dtype)          # what callers see (Flux.dtype: forward() returns this)
    pdtype = engine_flow_dtype(config.flow_dtype)  # what the parameters are held in: bf16, whatever the config asks for
    with torch.device("meta"):
        model = Flux(config, dtype=dtype)
        if not config.prequantized_flow:
            model.type(pdtype)
    if state_dict is None and config.ckpt_path is not None:
        from safetensors.torch import load_file as load_sft

        state_dict = load_sft(config.ckpt_path, device="cpu")
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=False, assign=True)
        if not config.prequantized_flow:
            model.type(pdtype)
    model.requires_grad_(False)
    return model


class LoadedModels(BaseModel):
    flow: object
    ae: object = None
    clip: object = None
    t5: object = None
    config: ModelSpec

    model_config = {"arbitrary_types_allowed": True, "use_enum_values": True}


def load_autoencoder(config: ModelSpec, state_dict=None):
    """reference util.py:269-296: AutoEncoder(config.ae_params) + weights from config.ae_path (a BFL `ae.sft`) on config.ae_device in
    bf16.  A decoder-only state dict is accepted (`ae.encoder_loaded = False`: encode() then refuses to run on random weights).
    Returns None when no weights are available (offline runs): FluxPipeline.generate then returns latents."""
    import os

    from modules.autoencoder import AutoEncoder

    if state_dict is None:
        path = getattr(config, "ae_path", None)
        if not path or not os.path.exists(path):
            return None
        from safetensors.torch import load_file as load_sft

        state_dict = load_sft(path, device="cpu")
    ae = AutoEncoder(config.ae_params)
    missing, unexpected = ae.load_state_dict(state_dict, strict=False)
    if any(k.startswith("decoder.") for k in missing):
        raise RuntimeError(f"autoencoder checkpoint is missing decoder weights: {[k for k in missing if k.startswith('decoder.')][:4]} ...")
    ae.encoder_loaded = not any(k.startswith("encoder.") for k in missing)
    return ae.to(device=into_device(config.ae_device), dtype=torch.bfloat16)


def load_text_encoders(config: ModelSpec, clip_kwargs=None, t5_kwargs=None):
    """reference util.py:262-280: (clip, t5) HFEmbedders -- CLIP at max_length 77, T5 at config.text_enc_max_length -- on
    config.text_enc_device.  `config.clip_path` / `text_enc_path` must be local HF-layout directories (no network here); when either
    is absent, or no `*_kwargs` offline hooks (hf_config / state_dict / tokenizer) are given, returns (None, None): the pipeline then
    expects pre-computed embeddings."""
    import os

    from modules.conditioner import HFEmbedder

    def have(path, kw):
        return kw is not None or (isinstance(path, str) and os.path.isdir(path))

    if not (have(config.clip_path, clip_kwargs) and have(config.text_enc_path, t5_kwargs)):
        return None, None
    dev = into_device(config.text_enc_device)
    clip = HFEmbedder(config.clip_path, max_length=77, torch_dtype=into_dtype(config.text_enc_dtype), device=dev, is_clip=True,
                      quantization_dtype=config.clip_quantization_dtype, **(clip_kwargs or {}))
    t5 = HFEmbedder(config.text_enc_path, max_length=config.text_enc_max_length, torch_dtype=into_dtype(config.text_enc_dtype), device=dev,
                    quantization_dtype=config.text_enc_quantization_dtype, **(t5_kwargs or {}))
    return clip, t5


def load_models_from_config(config: ModelSpec, state_dict=None, ae_state_dict=None, clip_kwargs=None, t5_kwargs=None) -> LoadedModels:
    """reference util.py:325-333."""
    clip, t5 = load_text_encoders(config, clip_kwargs, t5_kwargs)
    return LoadedModels(flow=load_flow_model(config, state_dict), ae=load_autoencoder(config, ae_state_dict), clip=clip, t5=t5, config=config)


def load_models_from_config_path(path: str) -> LoadedModels:
    return load_models_from_config(load_config_from_path(path))
