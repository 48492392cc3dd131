"""Generates tests/golden/g11_vae_full.safetensors: the UNMODIFIED reference VAE (/root/reference/modules/autoencoder.py, CPU) at the
REAL FLUX autoencoder geometry -- ch 128, ch_mult [1, 2, 4, 4], 2 res blocks, z_channels 16 (reference util.py:99-110) -- on a
256x256 image / 32x32 latent, decode and encode, fp32 and under torch.autocast(bf16) (what flux_pipeline.py:431-434 runs).
The 84 M weights are not stored: both sides rebuild them from a seed with vae_oracle.synth_state_dict (key names / shapes are the
BFL `ae.sft` layout, identical in the reference module and in flux-fp8-api_amd/modules/autoencoder.py).
    python oracle/gen_golden_vae_full.py        (needs /root/reference; the committed fixture is what the GPU test reads)"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: F401

sys.path.insert(0, "/root/reference")
from modules.autoencoder import AutoEncoder, AutoEncoderParams  # noqa: E402

import vae_oracle as vo  # noqa: E402


def main():
    from safetensors.torch import save_file

    P = vo.FULL_PARAMS
    ae = AutoEncoder(AutoEncoderParams(**P)).eval()
    sd = vo.synth_state_dict({k: v.shape for k, v in ae.state_dict().items()}, seed=7)
    ae.load_state_dict(sd, strict=True)
    z, x = vo.full_inputs()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    with torch.no_grad():
        ref32 = ae.decode(z)
        with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
            ref_ac = ae.decode(z)
        o32 = vo.decode(sd, P, z, autocast=False)
        # bit-equal at the small geometry of g8_vae; at this size torch picks other conv / SDPA blockings for the two call paths
        # (channels-last hints, batch-1 SDPA): the restatement must agree to fp32 rounding
        print(f"decode: oracle fp32 vs reference fp32 rel-L2 {rel(o32, ref32):.3e}, bit-equal {bool(torch.equal(o32, ref32))}")
        assert rel(o32, ref32) <= 1e-5
        m32 = ae.encoder(x)
        with torch.autocast("cpu", dtype=torch.bfloat16, cache_enabled=False):
            m_ac = ae.encoder(x)
        om32 = vo.encode_moments(sd, P, x, autocast=False)
        print(f"encode: oracle fp32 vs reference fp32 rel-L2 {rel(om32, m32):.3e}, bit-equal {bool(torch.equal(om32, m32))}")
        assert rel(om32, m32) <= 1e-5
    print(f"decode: reference autocast vs fp32 rel-L2 {rel(ref_ac, ref32):.3e}; encode moments: {rel(m_ac, m32):.3e}")
    out = {"dec_ref_fp32": ref32, "dec_ref_autocast": ref_ac.to(torch.bfloat16), "enc_moments_fp32": m32,
           "enc_moments_autocast": m_ac.to(torch.bfloat16)}
    path = os.path.join(HERE, "..", "tests", "golden", "g11_vae_full.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
