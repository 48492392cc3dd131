"""Generates tests/golden/g8_vae.safetensors: pins oracle/vae_oracle.py against the UNMODIFIED reference VAE
(/root/reference/modules/autoencoder.py, imported on CPU) and stores a small VAE (decoder + encoder) with latents / images and the reference's outputs.
    python oracle/gen_golden_vae.py        (needs /root/reference; the committed fixture is what the tests read)"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: F401  (stub loguru, torch.version.cuda)

sys.path.insert(0, "/root/reference")
from modules.autoencoder import AutoEncoder, AutoEncoderParams  # noqa: E402

import vae_oracle as vo  # noqa: E402

PARAMS = dict(resolution=32, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=4, scale_factor=0.3611,
              shift_factor=0.1159)


def main():
    from safetensors.torch import save_file

    torch.manual_seed(0)
    ae = AutoEncoder(AutoEncoderParams(**PARAMS)).eval()
    with torch.no_grad():
        for n, p in ae.named_parameters():
            if "norm" in n and n.endswith(".weight"):
                p.copy_(1 + 0.1 * torch.randn_like(p))
            elif "norm" in n and n.endswith(".bias"):
                p.copy_(0.1 * torch.randn_like(p))
        for n, p in ae.named_parameters():
            p.copy_(p.to(torch.bfloat16).float())  # ae_dtype = bfloat16 in the reference configs
    sd = {k: v.clone() for k, v in ae.state_dict().items()}
    z = torch.randn(2, 4, 16, 16)
    with torch.no_grad():
        ref32 = ae.decode(z)
        with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):  # flux_pipeline.py:431-434 (device = cpu here)
            ref_ac = ae.decode(z)
        o32 = vo.decode(sd, PARAMS, z, autocast=False)
        oac = vo.decode(sd, PARAMS, z, autocast=True)
    assert torch.equal(o32, ref32), "fp32 restatement must be bit-equal to the reference"
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    print(f"autocast: restatement vs reference(cpu autocast) rel-L2 {rel(oac, ref_ac):.3e}; vs fp32: reference {rel(ref_ac, ref32):.3e}, "
          f"restatement {rel(oac, ref32):.3e}")
    assert rel(oac, ref32) <= 1.5 * rel(ref_ac, ref32)
    # ---- encoder (img2img): moments, and the sampled latent with the reference's own randn_like draw ----
    x = (torch.rand(2, 3, 32, 32) * 2 - 1).to(torch.bfloat16).float()  # the pipeline hands the VAE a bf16 image (flux_pipeline.py:481-487)
    with torch.no_grad():
        m32 = ae.encoder(x)
        with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
            m_ac = ae.encoder(x)
        torch.manual_seed(123)
        e32 = ae.encode(x)
        torch.manual_seed(123)
        enc_noise = torch.randn_like(m32[:, :4])
        om32 = vo.encode_moments(sd, PARAMS, x, autocast=False)
        omac = vo.encode_moments(sd, PARAMS, x, autocast=True)
        oe32 = vo.encode(sd, PARAMS, x, noise=enc_noise, autocast=False)
    assert torch.equal(om32, m32), "fp32 encoder restatement must be bit-equal to the reference"
    assert torch.equal(oe32, e32), "fp32 encode (sampled with the same draw) must be bit-equal to the reference"
    print(f"encoder autocast: restatement vs reference(cpu autocast) rel-L2 {rel(omac, m_ac):.3e}; vs fp32: reference {rel(m_ac, m32):.3e}, "
          f"restatement {rel(omac, m32):.3e}")
    assert rel(omac, m32) <= 1.5 * rel(m_ac, m32)
    out = {"sd." + k: v.to(torch.bfloat16) for k, v in sd.items()}
    out.update({"z": z, "ref_fp32": ref32, "ref_autocast": ref_ac.float(), "oracle_autocast": oac.float()})
    out.update({"enc_x": x, "enc_noise": enc_noise, "enc_moments_fp32": m32, "enc_moments_autocast": m_ac.float(),
                "enc_oracle_moments_autocast": omac.float(), "enc_encode_fp32": e32})
    save_file(out, os.path.join(HERE, "..", "tests", "golden", "g8_vae.safetensors"))


if __name__ == "__main__":
    main()
