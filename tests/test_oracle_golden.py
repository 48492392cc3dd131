"""CPU: the oracle reproduces the fixtures that oracle/gen_golden.py produced from the UNMODIFIED reference.

In the build container these comparisons are bit-exact (asserted by gen_golden.py itself).  The model-level tensors go
through torch's CPU GEMM / SDPA kernels, whose summation order depends on the host ISA (AMX vs AVX-512), so on other
hosts they are compared with a tolerance; everything that is integer-like (fp8 bytes of casts, scales, trial counters,
schedules, LoRA re-quantised bytes up to rare rounding flips) is compared exactly.
"""
import os

import pytest
import torch
from safetensors.torch import load_file

import flux_oracle as fo
from parity_util import assert_f8_close

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return load_file(os.path.join(G, name + ".safetensors"))


def close_or_equal(a, b, rel, what):
    if torch.equal(a, b):
        return
    err = ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert err <= rel, f"{what}: rel-L2 {err:.3e} > {rel}"


def tiny_params(schnell=False):
    return fo.FluxParams(hidden_size=256, num_heads=2, depth=2, depth_single_blocks=2, context_in_dim=128, vec_in_dim=64,
                         guidance_embed=not schnell)


def test_g1_cast_tables_exact():
    g = load("g1_casts")
    x = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    ok = ~torch.isnan(x)
    one = torch.tensor(1.0)
    for name, dt, mx in (("e5m2", torch.float8_e5m2, 57344.0), ("e4m3", torch.float8_e4m3fn, 448.0)):
        mine = fo.to_fp8_saturated(x, one, mx).to(dt).view(torch.uint8)
        assert torch.equal(mine[ok], g[name][ok])
    # spot values from SURVEY.md §8c G1
    e4 = lambda v: torch.tensor(v).bfloat16().clamp(-448, 448).to(torch.float8_e4m3fn).float().item()
    e5 = lambda v: torch.tensor(v).bfloat16().clamp(-57344, 57344).to(torch.float8_e5m2).float().item()
    assert e4(2.0 ** -9) == 2.0 ** -9 and e4(2.0 ** -10) == 0.0 and e4(464.0) == 448.0 and e4(480.0) == 448.0
    assert e5(1.375) == 1.5 and e5(1.125) == 1.0 and e5(2.0 ** -17) == 0.0 and e5(61440.0) == 57344.0


@pytest.mark.parametrize("amax,e4,e5", [(0.0, 448.0, 57344.0), (1e-13, 448.0, 57344.0), (0.5, 448.0, 57344.0), (1.0, 448.0, 57344.0),
                                        (2.0, 224.0, 28672.0), (448.0, 1.0, 128.0)])
def test_g2_amax_to_scale_edges(amax, e4, e5):
    a = torch.tensor(amax, dtype=torch.float32)
    assert fo.amax_to_scale(a, 448.0).item() == e4 and fo.amax_to_scale(a, 57344.0).item() == e5


def test_g3_calibration_trace():
    g = load("g3_calibration")
    st = fo.F8LinearState(g["weight"], g["bias"])
    assert torch.equal(st.float8_data.view(torch.uint8), g["float8_data"]) and st.scale.item() == g["w_scale"].item()
    for call in range(15):
        tr = {}
        y = st(g[f"x{call}"], trace=tr, tag="l")
        s = g[f"state{call}"]
        assert st.input_scale.item() == s[0].item() and st.input_scale_reciprocal.item() == s[1].item()
        assert st.trial_index == int(s[2].item()) and float(st.input_scale_initialized) == s[3].item()
        assert torch.equal(tr["l.x8"].view(torch.uint8), g[f"x8_{call}"])
        close_or_equal(y, g[f"y{call}"], 2e-3, f"call {call}")
    assert torch.equal(st.input_amax_trials, g["trials"]) and st.input_scale_initialized


def test_g4_tables_and_schedule():
    g = load("g4_ops")
    img_ids, txt_ids = fo.make_ids(1, 64, 64, 512, torch.bfloat16)
    pe = fo.rope_table(torch.cat((txt_ids, img_ids), 1), [16, 56, 56], 10000, torch.bfloat16)
    assert torch.equal(pe[0, 0, :, :, 0, 0], g["pe_cos"]) and torch.equal(pe[0, 0, :, :, 1, 0], g["pe_sin"])
    assert torch.equal(pe[0, 0, :, :, 0, 1], -g["pe_sin"]) and torch.equal(pe[0, 0, :, :, 1, 1], g["pe_cos"])
    assert fo.get_schedule(28, 4096) == g["schedule_28_4096"].tolist()
    assert fo.get_schedule(12, 2304) == g["schedule_12_2304"].tolist()
    assert fo.get_schedule(4, 256, shift=False) == g["schedule_4_256_noshift"].tolist()
    ts = fo.get_schedule(28, 4096)
    assert ts[0] == 1.0 and abs(ts[1] - 0.98841) < 1e-5 and ts[-1] == 0.0
    te = fo.timestep_embedding(g["temb_in"], 256).bfloat16()
    assert torch.equal(te, g["temb"])


@pytest.mark.parametrize("qname,quant", [("bf16", None), ("fp8", dict(modulation=True, embedders=False)),
                                          ("fp8_emb", dict(modulation=True, embedders=True)), ("fp8_nomod", dict(modulation=False, embedders=False))])
def test_g5_model_forward(qname, quant):
    from fluxmi import synth

    g = load(f"g5_blocks_{qname}")
    p = tiny_params()
    sd = synth.make_state_dict(p, seed=0)
    orc = fo.FluxOracle(sd, p, quantize=quant)
    assert orc.n_f8() == {"bf16": 0, "fp8": 26, "fp8_emb": 34, "fp8_nomod": 20}[qname]  # SURVEY.md §8a row 7 (per 2+2 blocks)
    inp = synth.make_inputs(p, 64, 48, 24, batch=2, seed=3, real_tokens=8)
    for call in range(15):
        t = torch.full((2,), 1.0 - 0.06 * call, dtype=torch.bfloat16)
        gd = torch.full((2,), 3.5, dtype=torch.bfloat16)
        out = orc.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], gd)
        if call in (0, 12, 14):
            close_or_equal(out, g[f"pred{call}"], 1e-2 if quant is None else 6e-2, f"{qname} call {call}")
    if quant is not None:
        names = sorted(n for n, m in orc.lin.items() if isinstance(m, fo.F8LinearState))
        ws = torch.tensor([orc.lin[n].scale.item() for n in names])
        assert torch.equal(ws, g["weight_scales"])
        s = torch.tensor([orc.lin[n].input_scale.item() for n in names])
        assert ((s - g["input_scales"]).abs() <= 0.3 * g["input_scales"]).all()


@pytest.mark.parametrize("variant", ["dev_bf16", "dev_fp8", "schnell_bf16", "schnell_fp8"])
def test_g6_euler_loop(variant):
    from fluxmi import synth

    g = load(f"g6_loop_{variant}")
    schnell, q = variant.startswith("schnell"), variant.endswith("fp8")
    p = tiny_params(schnell)
    sd = synth.make_state_dict(p, seed=0)
    orc = fo.FluxOracle(sd, p, quantize=dict(modulation=True, embedders=False) if q else None)
    inp = synth.make_inputs(p, 64, 64, 32, batch=1, seed=7, real_tokens=8)
    ts = g["timesteps"].tolist()
    assert ts == fo.get_schedule(4 if schnell else 16, 16, shift=not schnell)
    lat = fo.denoise(orc, inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts, guidance=3.5)
    close_or_equal(lat, g["latents"], 6e-2 if q else 1e-2, variant)


def test_g7_lora_fuse_unfuse():
    from fluxmi import synth

    g = load("g7_lora")
    p = tiny_params()
    orc = fo.FluxOracle(synth.make_state_dict(p, seed=0), p, quantize=dict(modulation=True, embedders=False))
    lora = {k[len("lora."):]: (v if k.endswith("weight") else v.item()) for k, v in g.items() if k.startswith("lora.")}
    names = ["double_blocks.0.img_attn.qkv", "double_blocks.0.img_attn.proj", "single_blocks.1.linear2"]
    orc.fuse_lora(lora, 0.8)
    for n in names:
        assert abs(orc.lin[n].scale.item() - g[n + ".fused.scale"].item()) <= 1e-6 * g[n + ".fused.scale"].item()
        assert_f8_close(orc.lin[n].float8_data, g[n + ".fused.float8_data"].view(torch.float8_e4m3fn), 1, 0.999, n)
    orc.fuse_lora(lora, 0.8, sign=-1.0)
    for n in names:
        assert_f8_close(orc.lin[n].float8_data, g[n + ".unfused.float8_data"].view(torch.float8_e4m3fn), 1, 0.999, n)


def test_pack_unpack_round_trip_and_ids():
    x = torch.randn(2, 16, 12, 8)
    t = fo.pack_latent(x)
    assert t.shape == (2, 24, 64)
    assert torch.equal(fo.unpack_latent(t, 96, 64), x)
    # channel order (c, ph, pw): token 0 = [x[:, c, 0:2, 0:2] flattened per c]
    assert torch.equal(t[0, 0].reshape(16, 2, 2), x[0, :, 0:2, 0:2])
    img_ids, txt_ids = fo.make_ids(2, 6, 4, 5, torch.bfloat16)
    assert img_ids.shape == (2, 24, 3) and txt_ids.shape == (2, 5, 3) and not txt_ids.any()
    assert img_ids[0, 5].tolist() == [0.0, 1.0, 1.0]  # row-major (row, col)


def test_g8_vae_decoder_oracle_matches_reference_fixture():
    """oracle/vae_oracle.py (restatement of modules/autoencoder.py:203-283,330-332) vs the outputs of the
    unmodified reference stored by oracle/gen_golden_vae.py: fp32 bit-equal; the autocast(bf16) restatement reproducible."""
    import vae_oracle as vo
    from safetensors.torch import load_file

    g = load("g8_vae")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    P = dict(ch_mult=[1, 2], num_res_blocks=1, scale_factor=0.3611, shift_factor=0.1159)
    with torch.no_grad():
        o32 = vo.decode(sd, P, g["z"], autocast=False)
        oac = vo.decode(sd, P, g["z"], autocast=True)
    assert torch.equal(o32, g["ref_fp32"])
    assert torch.equal(oac.float(), g["oracle_autocast"])
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(oac, g["ref_fp32"]) <= 1.5 * rel(g["ref_autocast"], g["ref_fp32"])


def test_g8_vae_encoder_oracle_matches_reference_fixture():
    """oracle/vae_oracle.py encode_moments / encode (restatement of modules/autoencoder.py:95-107,
    123-200,286-299,326-329) vs the unmodified reference: fp32 moments and the sampled latent (same randn draw) bit-equal."""
    import vae_oracle as vo

    g = load("g8_vae")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    P = dict(ch_mult=[1, 2], num_res_blocks=1, scale_factor=0.3611, shift_factor=0.1159)
    with torch.no_grad():
        m32 = vo.encode_moments(sd, P, g["enc_x"], autocast=False)
        mac = vo.encode_moments(sd, P, g["enc_x"], autocast=True)
        e32 = vo.encode(sd, P, g["enc_x"], noise=g["enc_noise"], autocast=False)
    assert m32.shape == (2, 8, 16, 16)
    assert torch.equal(m32, g["enc_moments_fp32"])
    assert torch.equal(e32, g["enc_encode_fp32"])
    assert torch.equal(mac.float(), g["enc_oracle_moments_autocast"])
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    assert rel(mac, g["enc_moments_fp32"]) <= 1.5 * rel(g["enc_moments_autocast"], g["enc_moments_fp32"])
