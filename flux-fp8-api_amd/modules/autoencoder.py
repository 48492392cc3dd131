"""VAE parameters of the reference config schema (modules/autoencoder.py:11-21 of aredden/flux-fp8-api).

Only the pydantic parameter block is needed by the denoise hot path (every config JSON carries
`ae_params`).  The VAE encode/decode itself is row 1 of SURVEY.md §8(f) "next": not built yet."""
from pydantic import BaseModel


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


class AutoEncoder:  # placeholder so `from modules.autoencoder import AutoEncoder` keeps importing
    def __init__(self, params: AutoEncoderParams):
        raise NotImplementedError("fluxmi: the VAE is outside the denoise hot path (SURVEY.md §8f row 1); "
                                  "FluxPipeline returns latents when no autoencoder is attached")
