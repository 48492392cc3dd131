"""FluxPipeline on MI355X -- the pipeline surface of the reference (flux_pipeline.py of aredden/flux-fp8-api)
over the native fluxmi denoise engine.

Hot path kept here: seed -> noise -> 2x2 patch packing + position ids ->
shifted schedule -> the Euler denoise loop (natively, hipGraph-replayed) -> unpack.
Same public names / kwargs / defaults as the reference: FluxPipeline.load_pipeline_from_config_path /
load_pipeline_from_config / generate / load_lora / unload_lora / compile / set_seed / get_schedule /
get_noise / prepare / unpack / vae_decode / into_bytes / load_init_image_if_needed / resize_center_crop / preprocess_latent.

SURVEY.md §8f rows 1-2 are built around it: the native VAE decoder (latents -> JPEG) and encoder (img2img: `init_image`, `strength`),
and the text conditioning (flux_emphasis.py prompt weighting over the native T5 / CLIP encoders of modules/conditioner.py) when
`config.text_enc_path` / `clip_path` point at local HF-layout directories.  Without them `generate()` takes the conditioning as
pre-computed embeddings (`prompt={"txt": [B,Lt,4096], "vec": [B,768]}`); it returns latents when no autoencoder is attached.  CPU-offload flags are accepted
and ignored (meaningless with 288 GB of HBM).  `compile()` keeps the reference's warm-up/calibration protocol
(flux_pipeline.py:197-212) but never calls torch.compile: the fused kernels + hipGraph replace it.
"""
from __future__ import annotations

import io
import math
import random
from typing import Callable, List, Optional, Union

import numpy as np
import torch

import lora_loading  # noqa: F401  (same import side as the reference)
from fluxmi import dist as fdist
from util import ModelSpec, ModelVersion, engine_flow_dtype, into_device, into_dtype, load_config_from_path, load_models_from_config

MAX_RAND = 2**32 - 1


class FluxPipeline:
    def __init__(self, name: str, offload: bool = False, clip=None, t5=None, model=None, ae=None,
                 dtype: torch.dtype = torch.float16, verbose: bool = False, flux_device="cuda:0", ae_device="cuda:1",
                 clip_device="cuda:1", t5_device="cuda:1", config: ModelSpec = None, debug: bool = False):
        if config is None:
            raise ValueError("ModelSpec config is required!")
        self.debug, self.name, self.verbose, self.offload = debug, name, verbose, offload
        self.device_flux = into_device(flux_device)
        self.device_ae, self.device_clip, self.device_t5 = into_device(ae_device), into_device(clip_device), into_device(t5_device)
        self.dtype = into_dtype(dtype)
        self.clip, self.t5, self.model, self.ae = clip, t5, model, ae
        self.rng = torch.Generator(device="cpu")
        self.ae_dtype = torch.bfloat16
        self.config = config
        self.offload_text_encoder = config.offload_text_encoder
        self.offload_vae = config.offload_vae
        self.offload_flow = config.offload_flow
        self.model.to(self.device_flux)
        if config.compile_blocks or config.compile_extras:
            self.compile()

    # ---- seeds / noise / schedule (reference flux_pipeline.py:126-149, 314-371) -------------------------------
    def set_seed(self, seed: int | None = None, seed_globally: bool = False):
        if isinstance(seed, (int, float)):
            seed = int(abs(seed)) % MAX_RAND
        elif isinstance(seed, str):
            try:
                seed = abs(int(seed)) % MAX_RAND
            except Exception:
                seed = abs(self.rng.seed()) % MAX_RAND
        else:
            seed = abs(self.rng.seed()) % MAX_RAND
        generator = torch.Generator(self.device_flux).manual_seed(seed)
        if seed_globally:
            torch.cuda.manual_seed_all(seed)
            np.random.seed(seed)
            random.seed(seed)
        return generator, seed

    def time_shift(self, mu: float, sigma: float, t: torch.Tensor):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def get_lin_function(self, x1: float = 256, y1: float = 0.5, x2: float = 4096, y2: float = 1.15) -> Callable[[float], float]:
        m = (y2 - y1) / (x2 - x1)
        b = y1 - m * x1
        return lambda x: m * x + b

    def get_schedule(self, num_steps: int, image_seq_len: int, base_shift: float = 0.5, max_shift: float = 1.15,
                     shift: bool = True) -> list[float]:
        timesteps = torch.linspace(1, 0, num_steps + 1)
        if shift:
            mu = self.get_lin_function(y1=base_shift, y2=max_shift)(image_seq_len)
            timesteps = self.time_shift(mu, 1.0, timesteps)
        return timesteps.tolist()

    def get_noise(self, num_samples: int, height: int, width: int, generator: torch.Generator, dtype=None, device=None) -> torch.Tensor:
        device = self.device_flux if device is None else device
        dtype = self.dtype if dtype is None else dtype
        return torch.randn(num_samples, 16, 2 * math.ceil(height / 16), 2 * math.ceil(width / 16), device=device, dtype=dtype,
                           generator=generator, requires_grad=False)

    # ---- img2img entry (reference flux_pipeline.py:399-420, 450-523) ---------------------------------------------------
    def load_init_image_if_needed(self, init_image):
        """str (path or base64 / data-URL) | PIL.Image | np.ndarray -> uint8 HWC tensor; a torch.Tensor is returned as is."""
        from base64 import standard_b64decode

        from PIL import Image

        if isinstance(init_image, str):
            try:
                init_image = Image.open(init_image)
            except Exception:
                init_image = Image.open(io.BytesIO(standard_b64decode(init_image.split(",")[-1])))
            init_image = torch.from_numpy(np.array(init_image)).type(torch.uint8)
        elif isinstance(init_image, np.ndarray):
            init_image = torch.from_numpy(init_image).type(torch.uint8)
        elif isinstance(init_image, Image.Image):
            init_image = torch.from_numpy(np.array(init_image)).type(torch.uint8)
        return init_image

    @staticmethod
    def resize_center_crop(img: torch.Tensor, height: int, width: int) -> torch.Tensor:
        """[..., C, H, W] -> shorter edge resized to min(width, height), then centre-cropped (zero-padded where still smaller) to
        (height, width).  The reference calls torchvision's TF.resize / TF.center_crop (flux_pipeline.py:450-457); torchvision is not in
        this image, so the same arithmetic is restated: antialiased bilinear in fp32 (torchvision upcasts bf16), long edge =
        int(size * long / short), crop origin = int(round((H - h) / 2))."""
        size = min(width, height)
        H, W = img.shape[-2:]
        short, long_ = (W, H) if W <= H else (H, W)
        if short != size:
            new_long = int(size * long_ / short)
            nh, nw = (new_long, size) if W <= H else (size, new_long)
            lead = img.shape[:-3]
            r = torch.nn.functional.interpolate(img.reshape(-1, *img.shape[-3:]).float(), size=(nh, nw), mode="bilinear", align_corners=False,
                                                antialias=True)
            img = r.to(img.dtype).reshape(*lead, *r.shape[-3:])
            H, W = nh, nw
        if width > W or height > H:
            pl, pt = (width - W) // 2 if width > W else 0, (height - H) // 2 if height > H else 0
            pr, pb = (width - W + 1) // 2 if width > W else 0, (height - H + 1) // 2 if height > H else 0
            img = torch.nn.functional.pad(img, (pl, pr, pt, pb))
            H, W = img.shape[-2:]
        top, left = int(round((H - height) / 2.0)), int(round((W - width) / 2.0))
        return img[..., top:top + height, left:left + width]

    @torch.inference_mode()
    def preprocess_latent(self, init_image=None, height: int = 720, width: int = 1024, num_steps: int = 20, strength: float = 1.0,
                          generator: torch.Generator = None, num_images: int = 1, noise: Optional[torch.Tensor] = None):
        """reference flux_pipeline.py:459-523: noise + schedule; with an init image: VAE-encode it, start the schedule at
        t_idx = int((1 - strength) * num_steps) and blend x = t * noise + (1 - t) * latent."""
        if init_image is not None:
            if self.ae is None:
                raise RuntimeError("fluxmi: img2img needs an autoencoder (config.ae_path) -- none is attached")
            if isinstance(init_image, np.ndarray):
                init_image = torch.from_numpy(init_image)
            init_image = init_image.permute(2, 0, 1).contiguous().to(self.device_ae, dtype=self.ae_dtype).div(127.5).sub(1)[None, ...]
            init_image = self.resize_center_crop(init_image, height, width)
            init_image = self.ae.encode(init_image).to(dtype=self.dtype, device=self.device_flux).repeat(num_images, 1, 1, 1)
        x = self.get_noise(num_images, height, width, generator=generator) if noise is None else noise
        x = x.to(device=self.device_flux, dtype=self.dtype)
        timesteps = self.get_schedule(num_steps, x.shape[-1] * x.shape[-2] // 4, shift=(self.name != "flux-schnell"))
        if init_image is not None:
            t_idx = int((1 - strength) * num_steps)
            t = timesteps[t_idx]
            timesteps = timesteps[t_idx:]
            x = t * x + (1.0 - t) * init_image
        return x, timesteps

    # ---- packing / ids (reference flux_pipeline.py:267-292, 440-448) ----------------------------------------------
    @staticmethod
    def pack(img: torch.Tensor) -> torch.Tensor:
        b, c, h, w = img.shape
        return img.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // 2) * (w // 2), c * 4)

    def unpack(self, x: torch.Tensor, height: int, width: int) -> torch.Tensor:
        b = x.shape[0]
        h, w = math.ceil(height / 16), math.ceil(width / 16)
        return x.reshape(b, h, w, -1, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, -1, h * 2, w * 2)

    @staticmethod
    def make_img_ids(bs: int, h2: int, w2: int, device, dtype) -> torch.Tensor:
        ids = torch.zeros(h2, w2, 3, device=device, dtype=dtype)
        ids[..., 1] = ids[..., 1] + torch.arange(h2, device=device, dtype=dtype)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(w2, device=device, dtype=dtype)[None, :]
        return ids[None].repeat(bs, 1, 1, 1).flatten(1, 2)

    @torch.inference_mode()
    def prepare(self, img: torch.Tensor, prompt, target_device=None, target_dtype=None):
        """-> img tokens, img_ids, vec, txt, txt_ids (reference flux_pipeline.py:234-312)."""
        target_device = self.device_flux if target_device is None else target_device
        target_dtype = self.dtype if target_dtype is None else target_dtype
        bs, c, h, w = img.shape
        tokens = self.pack(img)
        assert tokens.shape == (bs, (h // 2) * (w // 2), c * 4), f"{tokens.shape} != {(bs, (h // 2) * (w // 2), c * 4)}"
        prompts = None
        if isinstance(prompt, (list, tuple)):
            # reference flux_pipeline.py:267-278: a list with num_images == 1 sizes the batch by its length (the one noise sample is
            # repeated).  The reference then hands the LIST to parse_prompt_attention, whose regex scan raises TypeError; here every
            # prompt of the list is embedded on its own, which is what the batch sizing implies.
            if not prompt or not all(isinstance(q, str) for q in prompt):
                raise TypeError("fluxmi: a prompt list must be a non-empty list of str")
            prompts = list(prompt)
            if bs == 1 and len(prompts) > 1:
                bs = len(prompts)
                tokens = tokens.repeat_interleave(bs, dim=0)
            elif len(prompts) == 1:
                prompts = prompts * bs
            elif len(prompts) != bs:
                raise ValueError(f"fluxmi: {len(prompts)} prompts for a batch of {bs} images")
        img_ids = self.make_img_ids(bs, h // 2, w // 2, target_device, target_dtype)
        if isinstance(prompt, dict):
            txt = prompt["txt"].to(device=target_device, dtype=target_dtype)
            vec = prompt["vec"].to(device=target_device, dtype=target_dtype)
            if txt.shape[0] == 1 and bs > 1:
                txt, vec = txt.expand(bs, -1, -1), vec.expand(bs, -1)
        elif self.t5 is not None and self.clip is not None:
            from flux_emphasis import get_weighted_text_embeddings_flux

            kw = dict(device=self.device_clip, target_device=target_device, target_dtype=target_dtype, debug=self.debug)
            if prompts is not None:
                parts = [get_weighted_text_embeddings_flux(self, q, num_images_per_prompt=1, **kw) for q in prompts]
                vec, txt, txt_ids = (torch.cat([pt[i] for pt in parts], 0) for i in range(3))
                return tokens, img_ids, vec, txt, txt_ids
            if not isinstance(prompt, str):
                raise TypeError("fluxmi: prompt must be a str, a list of str, or a dict of pre-computed embeddings")
            vec, txt, txt_ids = get_weighted_text_embeddings_flux(self, prompt, num_images_per_prompt=bs, **kw)
            return tokens, img_ids, vec, txt, txt_ids
        else:
            raise RuntimeError("fluxmi: no text encoders attached (config.text_enc_path / clip_path not found). Pass "
                               "prompt={'txt': T5 states [B,Lt,4096], 'vec': CLIP pooled [B,768]} or load the encoders.")
        txt_ids = torch.zeros(bs, txt.shape[1], 3, device=target_device, dtype=target_dtype)
        return tokens, img_ids, vec, txt, txt_ids

    # ---- LoRA (reference flux_pipeline.py:151-177) -------------------------------------------------------------------
    def load_lora(self, lora_path, scale: float, name: Optional[str] = None):
        self.model.load_lora(path=lora_path, scale=scale, name=name)

    def unload_lora(self, path_or_identifier: str):
        self.model.unload_lora(path_or_identifier=path_or_identifier)

    # ---- calibration warm-up (the part of reference compile() that matters, flux_pipeline.py:197-212) -----------------
    @torch.inference_mode()
    def compile(self, prompt=None):
        if self.config.prequantized_flow:
            return
        p = self.model.params
        lt = self.config.text_enc_max_length
        if prompt is None:
            if self.t5 is not None and self.clip is not None:
                # the reference's own warm-up prompt (flux_pipeline.py:199): the frozen input scales then reflect real text activations
                prompt = "A beautiful test image used to solidify the fp8 nn.Linear input scales prior to compilation 😉"
            else:
                # no encoders attached (offline runs): synthetic embeddings of T5 / CLIP-like statistics.  The scales are then tuned on
                # noise-like conditioning -- callers with real embeddings should pass them as `prompt`
                g = torch.Generator().manual_seed(10)
                prompt = {"txt": 0.1 * torch.randn(1, lt, p.context_in_dim, generator=g), "vec": torch.randn(1, p.vec_in_dim, generator=g)}
        # batch-sharded replicas: every layer's running amax is MAX-reduced across the ranks inside each calibrating step, so all
        # replicas freeze the SAME input scales (float8_quantize.py:227 takes amax over the whole batch)
        world = fdist.world_size()
        if world > 1:
            self.model.enable_amax_exchange()
        kw = dict(prompt=prompt, height=768, width=768, num_steps=12, guidance=3.5, seed=10, silent=True, output_type="latent",
                  num_images=max(1, world))
        if self.name == ModelVersion.flux_schnell.value or self.name == "flux-schnell":
            kw["num_steps"] = 4
            for _ in range(3):
                self.generate(**kw)
        else:
            self.generate(**kw)
            self.generate(**{**kw, "num_steps": 1})  # 13th call: freezes the input scales
        if world > 1:
            self.model.enable_amax_exchange(False)

    # ---- the request (reference flux_pipeline.py:526-663) ---------------------------------------------------------------
    @torch.inference_mode()
    def generate(self, prompt, width: int = 720, height: int = 1024, num_steps: int = 24, guidance: float = 3.5,
                 seed: int | None = None, init_image=None, strength: float = 1.0, silent: bool = False, num_images: int = 1,
                 return_seed: bool = False, jpeg_quality: int = 99, output_type: str = "jpeg", noise: Optional[torch.Tensor] = None,
                 use_graph: bool = True):
        num_steps = 4 if self.name == "flux-schnell" else num_steps
        init_image = self.load_init_image_if_needed(init_image) if init_image is not None else None
        height, width = 16 * (height // 16), 16 * (width // 16)
        generator, seed = self.set_seed(seed)
        world, rank = fdist.world_size(), fdist.rank()
        cal = getattr(self.model, "calibration_state", None)
        # the batch prepare() will build: a list prompt with one noise sample sizes the batch by its length (flux_pipeline.py:267-278)
        eff_images = len(prompt) if (isinstance(prompt, (list, tuple)) and num_images == 1) else num_images
        if world > 1 and eff_images < world and cal is not None and cal()[0] is False:
            # every rank sees the same (batch, world, calibration state), so every rank raises -- before any collective.  A rank
            # with an empty shard would skip the calibrating steps: its trial counters would not advance and (with or without the in-step
            # amax exchange) the replicas would freeze different input scales, silently.
            raise RuntimeError(f"fluxmi: calibrating with fewer images ({eff_images}) than ranks ({world}); use num_images >= world_size "
                               "until the F8Linear input scales are frozen (FluxPipeline.compile() does)")
        # batch-sharded replicas: every rank denoises its own slice of the batch (rank 0's noise / init latent is
        # what the broadcast below distributes)
        noise, timesteps = self.preprocess_latent(init_image=init_image, height=height, width=width, num_steps=num_steps, strength=strength,
                                                  generator=generator, num_images=num_images, noise=noise)
        img, img_ids, vec, txt, txt_ids = map(lambda x: x.contiguous(), self.prepare(noise, prompt))
        num_images = img.shape[0]  # a list prompt with num_images == 1 sizes the batch (prepare)
        if world > 1:
            txt, vec, img = fdist.broadcast_request(txt, vec, img, src=0)
            lo, hi = fdist.shard_bounds(img.shape[0], rank, world)
            img, img_ids, vec, txt, txt_ids = (t[lo:hi].contiguous() for t in (img, img_ids, vec, txt, txt_ids))
        if img.shape[0] == 0:
            # more ranks than images (frozen scales only, see above): this rank has nothing to denoise but still takes part in the gather
            # below (an exception or an early return here would leave the other ranks blocked in the collective)
            latents = img.new_empty((0,) + tuple(img.shape[1:]))
        else:
            latents = self.model.denoise(img, img_ids, txt, txt_ids, vec, timesteps, guidance=guidance, use_graph=use_graph)
        if world > 1:
            latents = fdist.gather_latents(latents, num_images, dst=0)
            if latents is None:  # only the gather rank decodes / returns the images
                return (None, seed) if return_seed else None
        if output_type == "latent" or self.ae is None:
            out = self.unpack(latents.float(), height, width) if latents is not None else None
            return (out, seed) if return_seed else out
        img_px = self.vae_decode(latents, height, width)
        # output_type "uint8": the [B, H, W, 3] array into_bytes hands to the JPEG encoder (the pixel-parity tests compare it)
        out = self.to_uint8(img_px) if output_type == "uint8" else self.into_bytes(img_px, jpeg_quality=jpeg_quality)
        return (out, seed) if return_seed else out

    def vae_decode(self, x: torch.Tensor, height: int, width: int) -> torch.Tensor:
        x = self.unpack(x.to(self.device_ae).float(), height, width)
        with torch.autocast(device_type=self.device_ae.type, dtype=torch.bfloat16, cache_enabled=False):
            return self.ae.decode(x)

    @staticmethod
    def to_uint8(x: torch.Tensor) -> torch.Tensor:
        """[B, 3, H, W] in [-1, 1] -> [B, H, W, 3] uint8 on the host (reference flux_pipeline.py:385-393)"""
        return x.clamp(-1, 1).add(1.0).mul(127.5).clamp(0, 255).permute(0, 2, 3, 1).contiguous().to(torch.uint8).cpu()

    def into_bytes(self, x: torch.Tensor, jpeg_quality: int = 99) -> io.BytesIO:
        from PIL import Image

        px = self.to_uint8(x)
        imgs = [px[i].numpy() for i in range(px.shape[0])]
        im = imgs[0] if len(imgs) == 1 else np.vstack(imgs)
        buf = io.BytesIO()
        Image.fromarray(im).save(buf, format="JPEG", quality=jpeg_quality)
        buf.seek(0)
        return buf

    # ---- loading (reference flux_pipeline.py:665-729) -------------------------------------------------------------------------
    @classmethod
    def load_pipeline_from_config_path(cls, path: str, flow_model_path: str = None, debug: bool = False, **kwargs) -> "FluxPipeline":
        with torch.inference_mode():
            config = load_config_from_path(path)
            if flow_model_path:
                config.ckpt_path = flow_model_path
            extra = {k: kwargs.pop(k, None) for k in ("state_dict", "ae_state_dict", "clip_kwargs", "t5_kwargs")}
            for k, v in kwargs.items():
                if hasattr(config, k):
                    setattr(config, k, v)
            return cls.load_pipeline_from_config(config, debug=debug, **extra)

    @classmethod
    def load_pipeline_from_config(cls, config: ModelSpec, debug: bool = False, state_dict=None, ae_state_dict=None, clip_kwargs=None,
                                  t5_kwargs=None) -> "FluxPipeline":
        """`state_dict` / `ae_state_dict` / `clip_kwargs` / `t5_kwargs` are offline hooks (weights, HF configs and tokenizers handed over in
        memory instead of read from the config's paths)."""
        from float8_quantize import quantize_flow_transformer_and_dispatch_float8

        with torch.inference_mode():
            models = load_models_from_config(config, state_dict=state_dict, ae_state_dict=ae_state_dict, clip_kwargs=clip_kwargs,
                                             t5_kwargs=t5_kwargs)
            config = models.config
            flux_device = into_device(config.flux_device)
            # every shipped reference JSON says flow_dtype float16: accepted -- the engine computes in bf16 and the pipeline keeps the
            # configured dtype for what crosses its boundary (util.engine_flow_dtype; warned once)
            flux_dtype = into_dtype(config.flow_dtype)
            flow_model = models.flow
            if not config.prequantized_flow:
                flow_model = quantize_flow_transformer_and_dispatch_float8(
                    flow_model, flux_device, offload_flow=config.offload_flow, swap_linears_with_cublaslinear=False,
                    flow_dtype=engine_flow_dtype(config.flow_dtype), quantize_modulation=config.quantize_modulation,
                    quantize_flow_embedder_layers=config.quantize_flow_embedder_layers)
            else:
                flow_model.eval().requires_grad_(False)
        return cls(name=config.version, clip=models.clip, t5=models.t5, model=flow_model, ae=models.ae, dtype=flux_dtype, verbose=False,
                   flux_device=flux_device, ae_device=into_device(config.ae_device), clip_device=into_device(config.text_enc_device),
                   t5_device=into_device(config.text_enc_device), config=config, debug=debug)
