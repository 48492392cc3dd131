#!/usr/bin/env python
"""Pins the oracle against the UNMODIFIED reference at Flux-dev's REAL geometry and writes tests/golden/g10_full_<case>.safetensors.

Run in the build container only (needs /root/reference):   python oracle/gen_golden_full.py [case ...]
Cases / protocol: oracle/full_geometry.py.  For every case the reference model (modules/flux_model.py:506-716 with
float8_quantize.quantize_flow_transformer_and_dispatch_float8 applied, CPU, torch's own _scaled_mm / SDPA kernels) and the oracle
run the same two calls; forward hooks on every reference F8Linear and on every block give the reference's intermediates.
Asserted here, bit for bit: both predictions, the output of EVERY F8Linear, every block output, every input / weight scale.
The fixture stores samples + checksums of the ORACLE's trace (== the reference wherever the reference exposes the tensor).
"""
import os
import sys
import time

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
import full_geometry as fg
import synth

OUT = os.path.join(ROOT, "tests", "golden")


def build_ref(p, sd, quant, schnell=False):
    cfg = rutil.load_config(rutil.ModelVersion.flux_schnell if schnell else rutil.ModelVersion.flux_dev, flow_dtype="bfloat16")
    cfg.params.depth, cfg.params.depth_single_blocks = p.depth, p.depth_single_blocks
    assert cfg.params.guidance_embed == p.guidance_embed
    with torch.device("meta"):
        m = fm.Flux(cfg, dtype=torch.bfloat16)
        m.type(torch.bfloat16)
    m.load_state_dict(sd, strict=True, assign=True)
    m.eval()
    if quant is not None:  # (quant None = the reference's bf16 flow: plain nn.Linear everywhere)
        f8q.quantize_flow_transformer_and_dispatch_float8(
            m, torch.device("cpu"), flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
            quantize_modulation=quant["modulation"], quantize_flow_embedder_layers=quant["embedders"])
    return m


def ref_loop(ref, inp, timesteps, guidance):
    """the denoise loop of FluxPipeline.generate, flux_pipeline.py:619-651, around the unmodified model"""
    img = inp["img"]
    guidance_vec = torch.full((img.shape[0],), guidance, dtype=torch.bfloat16)
    t_vec = None
    for t_curr, t_prev in zip(timesteps[:-1], timesteps[1:]):
        if t_vec is None:
            t_vec = torch.full((img.shape[0],), t_curr, dtype=torch.bfloat16)
        else:
            t_vec = t_vec.reshape((img.shape[0],)).fill_(t_curr)
        pred = ref.forward(img=img, img_ids=inp["img_ids"], txt=inp["txt"], txt_ids=inp["txt_ids"], y=inp["y"], timesteps=t_vec,
                           guidance=guidance_vec)
        img = img + (t_prev - t_curr) * pred
    return img


@torch.inference_mode()
def run_case(name):
    t_all = time.time()
    case, p, sd, inp = fg.make_case(name, synth)
    print(f"[{name}] synthetic checkpoint: {sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters ({time.time() - t_all:.0f} s)", flush=True)
    ref = build_ref(p, {k: v for k, v in sd.items()}, case["quant"], schnell=case.get("schnell", False))
    print(f"[{name}] reference built + quantised ({time.time() - t_all:.0f} s)", flush=True)
    got = {}

    def lin_hook(nm):
        def h(mod, args, out):
            if recording[0] and (case["trace"] == "all"):
                got[nm + ".out"] = out.reshape(-1, out.shape[-1])
        return h

    def blk_hook(nm, double):
        def h(mod, args, out):
            if not recording[0]:
                return
            if double:
                got[nm + ".img_out"], got[nm + ".txt_out"] = out
            else:
                got[nm + ".out"] = out
        return h

    recording = [False]
    for nm, mod in ref.named_modules():
        if isinstance(mod, f8q.F8Linear):
            mod.register_forward_hook(lin_hook(nm))
    for i, b in enumerate(ref.double_blocks):
        b.register_forward_hook(blk_hook(f"double_blocks.{i}", True))
    for i, b in enumerate(ref.single_blocks):
        b.register_forward_hook(blk_hook(f"single_blocks.{i}", False))

    t0 = time.time()
    r0 = ref(*fg.call_args(inp, fg.T_CALIB))
    print(f"[{name}] reference calibrating call {time.time() - t0:.0f} s", flush=True)
    for mod in ref.modules():
        if isinstance(mod, f8q.F8Linear):
            mod.input_scale_initialized = True  # freeze after one trial (float8_quantize.py:273: forward now bypasses quantize_input)
    if case.get("lora"):
        t0 = time.time()
        ref.load_lora({k: v.clone() for k, v in fg.make_lora(p, **case["lora"]).items()}, 1.0, name="c5")  # Flux.load_lora -> apply_lora_to_model
        print(f"[{name}] reference LoRA fused into the fp8 weights ({time.time() - t0:.0f} s)", flush=True)
    recording[0] = case["trace"] != "none"
    t0 = time.time()
    r1 = ref(*fg.call_args(inp, fg.T_FROZEN))
    print(f"[{name}] reference frozen call {time.time() - t0:.0f} s, {len(got)} intermediates hooked", flush=True)
    recording[0] = False
    r_loop = None
    if case.get("loop_steps"):
        t0 = time.time()
        r_loop = ref_loop(ref, inp, fg.loop_schedule(case), fg.GUIDANCE)
        print(f"[{name}] reference {case['loop_steps']}-step Euler loop {time.time() - t0:.0f} s", flush=True)

    orc, o0, o1, tr = fg.run_oracle(name, p, sd, inp, log=lambda m: print(m, flush=True))
    assert torch.equal(r0, o0), f"{name}: calibrating prediction differs"
    assert torch.equal(r1, o1), f"{name}: frozen prediction differs"
    n_lin = 0
    for nm, st in orc.lin.items():
        if isinstance(st, fo.F8LinearState):
            rm = ref.get_submodule(nm)
            assert rm.input_scale.item() == st.input_scale.item() and rm.scale.item() == st.scale.item(), nm
            assert torch.equal(rm.float8_data.view(torch.uint8), st.float8_data.view(torch.uint8)), nm
            n_lin += 1
    missing = [k for k in got if k not in tr]
    assert not missing, missing[:5]
    for k, v in got.items():
        assert torch.equal(v.reshape(tr[k].shape), tr[k]), f"{name}: {k} differs from the reference"
    if r_loop is not None:
        assert torch.equal(r_loop, tr["loop_latents"]), f"{name}: loop latents differ from the reference"
    print(f"[{name}] oracle == reference bit for bit: 2 predictions, {len(got)} intermediates, {n_lin} F8Linear states"
          + (" (fp8 weight bytes + scales AFTER the LoRA fuse)" if case.get("lora") else "") + (f", the latents after {case['loop_steps']} Euler steps" if r_loop is not None else ""),
          flush=True)
    tr = dict(tr)
    tr["pred_calib"], tr["pred_frozen"] = o0, o1
    fg.add_lora_weight_entries(tr, orc, p, case)  # the fused + re-quantised weights themselves go into the fixture (samples + checksums)
    names = sorted(n for n, m in orc.lin.items() if isinstance(m, fo.F8LinearState))
    dg = fg.digest(tr)
    dg["input_scales"] = torch.tensor([orc.lin[n].input_scale.item() for n in names], dtype=torch.float32)
    dg["weight_scales"] = torch.tensor([orc.lin[n].scale.item() for n in names], dtype=torch.float32)
    path = os.path.join(OUT, f"g10_full_{name}.safetensors")
    save_file({k: v.contiguous() for k, v in dg.items()}, path, metadata={"reference_hooked": str(len(got)), "torch": torch.__version__})
    print(f"[{name}] wrote {path}: {len(dg)} entries, {os.path.getsize(path) / 1024:.0f} KiB ({time.time() - t_all:.0f} s total)", flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(fg.CASES)
    print("pinning oracle/flux_oracle.py against", ref_shims.REFERENCE_ROOT, "at full geometry:", names, flush=True)
    for n in names:
        run_case(n)
    print("all oracle == reference assertions passed")
