#!/usr/bin/env python
"""Pins the oracle against the UNMODIFIED reference and writes tests/golden/*.safetensors.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
The reference ships no tests or golden vectors for this path, so the fixtures are outputs of the
reference itself (imported through oracle/ref_shims.py, executed on CPU with torch's own kernels) on seeded
synthetic inputs.  Every fixture is first compared BIT-FOR-BIT with oracle/flux_oracle.py (assert), then stored;
tests/test_oracle_golden.py re-checks the oracle against the stored tensors on any machine.

Fixture index:
  g1_casts        exhaustive bf16 -> e5m2 / e4m3fn cast tables on the clamped domain, GELU(tanh)/SiLU tables
  g3_calibration  15-call trace of one reference F8Linear (running scale, trial_index, quantised bytes, outputs)
  g5_blocks_*     one Flux.forward of a 2+2-block model (hidden 256) for the 4 quantisation flag combos, calls 0/12/14
  g6_loop_*       3-step and 16-step Euler loops (bf16 and fp8) incl. calibration-phase steps
  g7_lora         float8_data / scale of qkv (uneven rank), proj (even rank, alpha) before/after fuse and after unfuse
  g4_ops          rope table, timestep embedding at the 28+1 schedule points, schedule values
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd", "fluxmi"))

import warnings

warnings.filterwarnings("ignore")
import torch
from safetensors.torch import save_file

import ref_shims

f8q, fm, rutil = ref_shims.import_reference()
import flux_oracle as fo
import synth  # fluxmi/synth.py imported as a plain module (the package __init__ is not needed here)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)


def tiny_cfg(schnell=False):
    cfg = rutil.load_config(rutil.ModelVersion.flux_schnell if schnell else rutil.ModelVersion.flux_dev, flow_dtype="bfloat16")
    p = cfg.params
    p.hidden_size, p.num_heads, p.depth, p.depth_single_blocks, p.context_in_dim, p.vec_in_dim = 256, 2, 2, 2, 128, 64
    return cfg


def build_ref(cfg, sd, quant):
    with torch.device("meta"):
        m = fm.Flux(cfg, dtype=torch.bfloat16)
        m.type(torch.bfloat16)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True, assign=True)
    m.eval()
    if quant is not None:
        f8q.quantize_flow_transformer_and_dispatch_float8(
            m, torch.device("cpu"), flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
            quantize_modulation=quant["modulation"], quantize_flow_embedder_layers=quant["embedders"])
    return m


def save(name, tensors):
    tensors = {k: v.contiguous() for k, v in tensors.items()}
    path = os.path.join(OUT, name + ".safetensors")
    save_file(tensors, path)
    print(f"  wrote {name}: {len(tensors)} tensors, {os.path.getsize(path) / 1024:.0f} KiB")


@torch.inference_mode()
def g1_casts():
    x = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    out = {}
    lin = f8q.F8Linear.from_linear(torch.nn.Linear(8, 8).bfloat16())
    one = torch.tensor(1.0)
    for name, dt, mx in (("e5m2", torch.float8_e5m2, 57344.0), ("e4m3", torch.float8_e4m3fn, 448.0)):
        ref = lin.to_fp8_saturated(x, one, mx).to(dt)
        mine = fo.to_fp8_saturated(x, one, mx).to(dt)
        ok = ~torch.isnan(x)
        assert torch.equal(ref.view(torch.uint8)[ok], mine.view(torch.uint8)[ok])
        out[name] = ref.view(torch.uint8)
    xf = torch.where(torch.isfinite(x), x, torch.zeros_like(x))
    out["gelu_tanh"] = torch.nn.functional.gelu(xf, approximate="tanh").view(torch.int16)
    out["silu"] = torch.nn.functional.silu(xf).view(torch.int16)
    save("g1_casts", out)


@torch.inference_mode()
def g3_calibration():
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(96, 64, generator=g) * 0.05).bfloat16()
    b = torch.randn(96, generator=g).bfloat16()
    lin = torch.nn.Linear(64, 96).bfloat16()
    lin.weight.data, lin.bias.data = w.clone(), b.clone()
    ref = f8q.F8Linear.from_linear(lin)
    st = fo.F8LinearState(w.clone(), b.clone())
    assert torch.equal(ref.float8_data.view(torch.uint8), st.float8_data.view(torch.uint8)) and ref.scale.item() == st.scale.item()
    out = {"weight": w, "bias": b, "float8_data": ref.float8_data.view(torch.uint8), "w_scale": ref.scale.reshape(1)}
    for call in range(15):
        x = (torch.randn(40, 64, generator=g) * (0.5 + 3.0 * ((call * 7) % 5))).bfloat16()
        if call == 3:
            x = x * 1e-3
        yr = ref(x)
        tr = {}
        yo = st(x, trace=tr, tag="l")
        assert torch.equal(yr, yo) and ref.input_scale.item() == st.input_scale.item() and ref.trial_index == st.trial_index
        assert ref.input_scale_initialized == st.input_scale_initialized
        out[f"x{call}"] = x
        out[f"x8_{call}"] = tr["l.x8"].view(torch.uint8)
        out[f"y{call}"] = yr
        out[f"state{call}"] = torch.tensor([ref.input_scale.item(), ref.input_scale_reciprocal.item(), float(ref.trial_index),
                                            float(ref.input_scale_initialized)], dtype=torch.float32)
    out["trials"] = ref.input_amax_trials.clone()
    save("g3_calibration", out)


QUANTS = {"bf16": None, "fp8": dict(modulation=True, embedders=False), "fp8_emb": dict(modulation=True, embedders=True),
          "fp8_nomod": dict(modulation=False, embedders=False)}


@torch.inference_mode()
def g5_blocks():
    cfg = tiny_cfg()
    p = cfg.params
    sd = synth.make_state_dict(p, seed=0)
    inp = synth.make_inputs(p, 64, 48, 24, batch=2, seed=3, real_tokens=8)
    for qname, quant in QUANTS.items():
        ref = build_ref(cfg, sd, quant)
        orc = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, fo.FluxParams(**p.model_dump()), quantize=quant)
        out = {}
        for call in range(15):
            t = torch.full((2,), 1.0 - 0.06 * call, dtype=torch.bfloat16)
            gd = torch.full((2,), 3.5, dtype=torch.bfloat16)
            a = ref(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], gd)
            b = orc.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], gd)
            assert torch.equal(a, b), (qname, call)
            if call in (0, 12, 14):
                out[f"pred{call}"] = a
        if quant is not None:
            names = sorted(n for n, m in orc.lin.items() if isinstance(m, fo.F8LinearState))
            for n in names:
                rm = ref.get_submodule(n)
                assert rm.input_scale.item() == orc.lin[n].input_scale.item()
            out["input_scales"] = torch.tensor([orc.lin[n].input_scale.item() for n in names], dtype=torch.float32)
            out["weight_scales"] = torch.tensor([orc.lin[n].scale.item() for n in names], dtype=torch.float32)
        save(f"g5_blocks_{qname}", out)


@torch.inference_mode()
def g6_loop():
    for schnell in (False, True):
        cfg = tiny_cfg(schnell)
        p = cfg.params
        sd = synth.make_state_dict(p, seed=0)
        inp = synth.make_inputs(p, 64, 64, 32, batch=1, seed=7, real_tokens=8)
        n = 4 if schnell else 16
        ts = fo.get_schedule(n, 16, shift=not schnell)
        for qname in ("bf16", "fp8"):
            ref = build_ref(cfg, sd, QUANTS[qname])
            orc = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, fo.FluxParams(**p.model_dump()), quantize=QUANTS[qname])
            # the reference loop (flux_pipeline.py:619-651) written out against the reference model
            img = inp["img"]
            gv = torch.full((1,), 3.5, dtype=torch.bfloat16)
            for t_curr, t_prev in zip(ts[:-1], ts[1:]):
                tv = torch.full((1,), t_curr, dtype=torch.bfloat16)
                pred = ref(img, inp["img_ids"], inp["txt"], inp["txt_ids"], tv, inp["y"], gv if not schnell else None)
                img = img + (t_prev - t_curr) * pred
            mine = fo.denoise(orc, inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts, guidance=3.5)
            assert torch.equal(img, mine), (schnell, qname)
            save(f"g6_loop_{'schnell' if schnell else 'dev'}_{qname}", {"latents": img, "timesteps": torch.tensor(ts, dtype=torch.float64)})


@torch.inference_mode()
def g7_lora():
    cfg = tiny_cfg()
    p = cfg.params
    sd = synth.make_state_dict(p, seed=0)
    ref = build_ref(cfg, sd, QUANTS["fp8"])
    orc = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, fo.FluxParams(**p.model_dump()), quantize=QUANTS["fp8"])
    g = torch.Generator().manual_seed(21)
    H = p.hidden_size
    lora = {
        "double_blocks.0.img_attn.qkv.lora_A.weight": torch.randn(3 * 16, H, generator=g) * 0.05,
        "double_blocks.0.img_attn.qkv.lora_B.weight": torch.randn(3 * H, 16, generator=g) * 0.05,
        "double_blocks.0.img_attn.proj.lora_A.weight": torch.randn(16, H, generator=g) * 0.05,
        "double_blocks.0.img_attn.proj.lora_B.weight": torch.randn(H, 16, generator=g) * 0.05,
        "double_blocks.0.img_attn.proj.alpha": torch.tensor(8.0),
        "single_blocks.1.linear2.lora_A.weight": torch.randn(16, 5 * H, generator=g) * 0.05,
        "single_blocks.1.linear2.lora_B.weight": torch.randn(H, 16, generator=g) * 0.05,
    }
    out = {"lora." + k: v for k, v in lora.items()}
    ref_lora = {k: (v.clone() if k.endswith("weight") else v.item()) for k, v in lora.items()}
    orc_lora = {k: (v.clone() if k.endswith("weight") else v.item()) for k, v in lora.items()}
    ref.load_lora(ref_lora, 0.8, name="g7")
    orc.fuse_lora(orc_lora, 0.8)
    names = ["double_blocks.0.img_attn.qkv", "double_blocks.0.img_attn.proj", "single_blocks.1.linear2"]
    for n in names:
        rm = ref.get_submodule(n)
        assert torch.equal(rm.float8_data.view(torch.uint8), orc.lin[n].float8_data.view(torch.uint8)), n
        assert rm.scale.item() == orc.lin[n].scale.item()
        out[n + ".fused.float8_data"] = rm.float8_data.view(torch.uint8)
        out[n + ".fused.scale"] = rm.scale.reshape(1)
    ref.unload_lora("g7")
    orc.fuse_lora(orc_lora, 0.8, sign=-1.0)
    for n in names:
        rm = ref.get_submodule(n)
        assert torch.equal(rm.float8_data.view(torch.uint8), orc.lin[n].float8_data.view(torch.uint8)), n
        out[n + ".unfused.float8_data"] = rm.float8_data.view(torch.uint8)
        out[n + ".unfused.scale"] = rm.scale.reshape(1)
    save("g7_lora", out)


@torch.inference_mode()
def g4_ops():
    cfg = tiny_cfg()
    out = {}
    img_ids, txt_ids = fo.make_ids(1, 64, 64, 512, torch.bfloat16)
    ids = torch.cat((txt_ids, img_ids), 1)
    emb = fm.EmbedND(128, 10000, [16, 56, 56], dtype=torch.bfloat16)
    pe_ref = emb(ids)
    assert torch.equal(pe_ref, fo.rope_table(ids, [16, 56, 56], 10000, torch.bfloat16))
    out["pe_cos"] = pe_ref[0, 0, :, :, 0, 0].contiguous()
    out["pe_sin"] = pe_ref[0, 0, :, :, 1, 0].contiguous()
    ts = fo.get_schedule(28, 4096)

    class P:  # the reference's schedule functions are methods; call them unbound (flux_pipeline.py is not importable: needs torchvision)
        pass

    t = torch.tensor(ts[:-1] + [3.5], dtype=torch.float32).bfloat16()
    te_ref = fm.timestep_embedding(t, 256)
    assert torch.equal(te_ref, fo.timestep_embedding(t, 256))
    out["schedule_28_4096"] = torch.tensor(ts, dtype=torch.float64)
    out["schedule_12_2304"] = torch.tensor(fo.get_schedule(12, 2304), dtype=torch.float64)
    out["schedule_4_256_noshift"] = torch.tensor(fo.get_schedule(4, 256, shift=False), dtype=torch.float64)
    out["temb_in"] = t
    out["temb"] = te_ref.bfloat16()
    save("g4_ops", out)


if __name__ == "__main__":
    print("pinning oracle/flux_oracle.py against", ref_shims.REFERENCE_ROOT)
    for fn in (g1_casts, g3_calibration, g4_ops, g5_blocks, g6_loop, g7_lora):
        print(fn.__name__)
        fn()
    print("all oracle == reference assertions passed")
