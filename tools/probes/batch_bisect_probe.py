"""Where does sample 0 of a batch of B first differ from the same sample alone?  Flux-dev width, 1 + 1 blocks, 1024^2: the engine is prepared by a
forward at batch 1 and at batch B, then the double block and the single block run stage by stage (fluxmi_engine_run_block) from the SAME x for sample 0;
after every stage sample 0's slice of every workspace buffer is compared between the two batch sizes.
    python tools/probes/batch_bisect_probe.py [--B 4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
sys.path.insert(0, ROOT)
import torch

import util
from bench import util_schedule
from float8_quantize import quantize_flow_transformer_and_dispatch_float8
from fluxmi import _lib, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
p = cfg.params
p.depth, p.depth_single_blocks = 1, 1
keys = ("img", "img_ids", "txt", "txt_ids", "y")
H, Hm, Lt = 3072, 12288, 512


def get(model, name, nbytes, offset=0):
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.call("fluxmi_engine_copy_buffer", model._engine, name.encode(), offset, ops._p(t), nbytes, 0, ops._stream())
    torch.cuda.synchronize()
    return t


def put(model, name, t, offset=0):
    t = t.contiguous()
    _lib.call("fluxmi_engine_copy_buffer", model._engine, name.encode(), offset, ops._p(t), t.numel() * t.element_size(), 1, ops._stream())


with torch.inference_mode():
    model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=2, device=dev))
    quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                  quantize_modulation=True, quantize_flow_embedder_layers=False)
    inp = {k: v.to(dev) for k, v in synth.make_inputs(p, 1024, 1024, 512, batch=a.B, seed=40).items()}
    Li = inp["img"].shape[1]
    L = Li + Lt
    sl = lambda n: tuple(inp[k][:n].contiguous() for k in keys)
    model.denoise(*sl(1), util_schedule(13, Li), guidance=3.5)
    assert model.calibration_state()[0]
    g = torch.Generator(device=dev).manual_seed(5)
    x0 = torch.randn(a.B, L, H, generator=g, device=dev).bfloat16()
    # sample-0 slices: (buffer, bytes per sample)
    per = {"x": L * H * 2, "a8": L * H, "attn8": L * H, "qkv": L * 3 * H * 2, "K": L * H * 2, "VT": H * ((L + 63) // 64 * 64) * 2, "h8": L * Hm, "cat8": L * (H + Hm),
           "pe": L * 64 * 2 * 2, "mod": None, "vec": H * 2, "txt_emb": Lt * H * 2}
    snaps = {}
    for Bb in (1, a.B):
        tv = torch.full((Bb,), 0.75, dtype=torch.bfloat16, device=dev)
        gv = torch.full((Bb,), 3.5, dtype=torch.bfloat16, device=dev)
        d = sl(Bb)
        pred = model(d[0], d[1], d[2], d[3], tv, d[4], gv, mode=1)
        torch.cuda.synchronize()
        snap = {"pred": pred[0].clone().view(torch.uint8).flatten()}
        mod_cols = (1 * 12 + 1 * 3 + 2) * H
        per["mod"] = mod_cols * 2
        for nm in ("mod", "vec", "pe", "txt_emb"):
            snap["after forward: " + nm] = get(model, nm, per[nm])
        for kind, nst, label in ((0, 8, "double"), (1, 5, "single")):
            put(model, "x", x0[:Bb])
            for st in range(nst):
                _lib.call("fluxmi_engine_run_block", model._engine, kind, 0, 1, st, st, ops._stream())
                torch.cuda.synchronize()
                for nm in ("x", "a8", "qkv", "K", "VT", "attn8", "h8", "cat8"):
                    snap[f"{label} stage {st}: {nm}"] = get(model, nm, per[nm])
        snaps[Bb] = snap
    first = None
    for k in snaps[1]:
        u, v = snaps[1][k], snaps[a.B][k]
        same = torch.equal(u, v)
        if not same or first is None or "after forward" in k or k == "pred":
            frac = (u != v).float().mean().item()
            print(f"{k:32s} {'identical' if same else 'DIFFERS in %.4f of its bytes' % frac}", flush=True)
        if not same and first is None:
            first = k
    print("first difference:", first)
