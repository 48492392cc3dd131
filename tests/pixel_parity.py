"""Prompt -> pixels parity chain (TEST INFRASTRUCTURE: used by tests/test_pixels_gpu.py and __graft_entry__.smoke()).

north_star: "outputs match the reference bf16 flow path on identical seeds/prompts within a stated per-pixel fp tolerance".  The two
chains compared here, on the same prompt, the same noise tensor (the seed's draw) and the same weights:

  engine : FluxPipeline.generate(prompt string): tokenizers -> native T5 / CLIP (text.hip) + prompt weighting -> fp8 (or bf16) flow on
           the HIP engine, hipGraph loop -> unpack -> native VAE decode (vae.hip) -> clamp / scale / uint8   [reference
           flux_pipeline.py:526-663: prepare :234-312, loop :619-651, vae_decode :423-438, into_bytes :373-421 up to the JPEG encoder]
  oracle : the reference's own fp32 text-conditioning outputs for that prompt (tests/golden/g9_text.safetensors, written by the
           UNMODIFIED reference through oracle/gen_golden_text.py) cast to the flow dtype -> oracle/flux_oracle.py (bf16 flow = the
           "reference bf16 flow path"; or its F8Linear restatement) -> oracle/vae_oracle.py under autocast(bf16) -> the same uint8 map

Pixels are compared BEFORE the JPEG encoder (libjpeg is a third-party codec on both sides; the uint8 array is what it is handed).
The tolerance is stated relative to what fp8 itself costs: d(engine-fp8, oracle-bf16) <= 1.25 x d(oracle-fp8, oracle-bf16), the pixel
form of SURVEY.md section 8(c) gate (iv); d = mean |delta| in 8-bit levels (PSNR is the same statement on the squared error).
"""
import math

import torch

import flux_oracle as fo
import vae_oracle as vo


def to_uint8(x: torch.Tensor) -> torch.Tensor:
    """reference flux_pipeline.py:373-421 (into_bytes) up to the JPEG encoder: [B,3,H,W] in [-1,1] -> [B,H,W,3] uint8.  The arithmetic runs in
    the decoder's output dtype (bf16 under the reference's autocast: add and mul each round to bf16; .type(uint8) truncates)."""
    return x.clamp(-1, 1).add(1.0).mul(127.5).clamp(0, 255).permute(0, 2, 3, 1).contiguous().to(torch.uint8).cpu()


def pixel_metrics(a: torch.Tensor, b: torch.Tensor) -> dict:
    """a, b: uint8 images of one shape -> per-pixel statistics in 8-bit levels (/255 units)"""
    assert a.shape == b.shape and a.dtype == torch.uint8 and b.dtype == torch.uint8
    d = (a.to(torch.int32) - b.to(torch.int32)).abs().double()
    mse = (d * d).mean().item()
    return dict(max_abs=int(d.max().item()), mean_abs=d.mean().item(), psnr_db=(10 * math.log10(255.0 ** 2 / mse) if mse > 0 else float("inf")),
                frac_equal=(d == 0).double().mean().item(), frac_within_2=(d <= 2).double().mean().item())


def fmt(m: dict) -> str:
    return (f"mean |d| {m['mean_abs']:.3f}/255, max |d| {m['max_abs']}/255, PSNR {m['psnr_db']:.1f} dB, identical {100 * m['frac_equal']:.1f} %, "
            f"within 2 levels {100 * m['frac_within_2']:.1f} %")


def oracle_pixels(sd, params: fo.FluxParams, quantize, ae_sd, ae_params: dict, txt, vec, noise_cal, noise, height, width, steps, guidance=3.5,
                  schnell=False):
    """the oracle chain from conditioning to uint8 pixels.  `quantize` None = the reference bf16 flow; a dict = its F8Linear flow, whose
    input scales are first calibrated exactly like the engine's in the test: one 13-step request on `noise_cal` (calls 1-12 collect
    amax, call 13 freezes: float8_quantize.py:220-246)."""
    orc = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, params, quantize=quantize)
    dt = torch.bfloat16
    txt, vec = txt.to(dt), vec.to(dt)
    B = noise.shape[0]
    h2, w2 = noise.shape[-2] // 2, noise.shape[-1] // 2
    img_ids, txt_ids = fo.make_ids(B, h2, w2, txt.shape[1], dt)
    if quantize is not None:
        fo.denoise(orc, fo.pack_latent(noise_cal.to(dt)), img_ids, txt, txt_ids, vec, fo.get_schedule(13, h2 * w2, shift=not schnell), guidance)
        assert all(m.input_scale_initialized for m in orc.lin.values() if isinstance(m, fo.F8LinearState))
    lat = fo.denoise(orc, fo.pack_latent(noise.to(dt)), img_ids, txt, txt_ids, vec, fo.get_schedule(steps, h2 * w2, shift=not schnell), guidance)
    z = fo.unpack_latent(lat.float(), height, width)  # flux_pipeline.py:428-430: unpack(x.float())
    return to_uint8(vo.decode(ae_sd, ae_params, z, autocast=True)), z  # z: what generate(output_type='latent') returns


def engine_pixels(pipe, prompt, noise_cal, noise, height, width, steps, guidance=3.5, calibrate=True):
    """the engine chain: one 13-step calibrating request on `noise_cal` (fp8 flows), then the request itself -> uint8 pixels + latents"""
    if calibrate:
        pipe.generate(prompt, width=width, height=height, num_steps=13, guidance=guidance, noise=noise_cal, output_type="latent", silent=True)
        ok, _ = pipe.model.calibration_state()
        assert ok, "13 calls must have frozen the input scales"
    px = pipe.generate(prompt, width=width, height=height, num_steps=steps, guidance=guidance, noise=noise, output_type="uint8", silent=True)
    lat = pipe.generate(prompt, width=width, height=height, num_steps=steps, guidance=guidance, noise=noise, output_type="latent", silent=True)
    return px, lat
