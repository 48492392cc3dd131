"""Is a sample's result independent of the batch it rides in?  Flux-dev width, 1 + 1 blocks, 1024^2: sample 0 alone vs inside a batch of B, over B and
over the kernel-selection knobs (which knob, switched off on BOTH sides, restores bit-equality tells which selection rule is batch-dependent).
    python tools/probes/batch_invariance_probe.py [--B 2,8,32] [--hw 1024,1024]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

import util
from bench import util_schedule
from float8_quantize import quantize_flow_transformer_and_dispatch_float8
from fluxmi import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", default="2,4,8,16,24,32")
ap.add_argument("--hw", default="1024,1024")
ap.add_argument("--depth", default="1,1")
a = ap.parse_args()
h, w = (int(v) for v in a.hw.split(","))
dev = torch.device("cuda:0")
cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
p = cfg.params
p.depth, p.depth_single_blocks = (int(v) for v in a.depth.split(","))
keys = ("img", "img_ids", "txt", "txt_ids", "y")
with torch.inference_mode():
    model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=2, device=dev))
    quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                  quantize_modulation=True, quantize_flow_embedder_layers=False)
    Bmax = max(int(v) for v in a.B.split(","))
    inp = {k: v.to(dev) for k, v in synth.make_inputs(p, h, w, 512, batch=Bmax, seed=40).items()}
    Li = inp["img"].shape[1]
    sl = lambda n: tuple(inp[k][:n].contiguous() for k in keys)
    model.denoise(*sl(1), util_schedule(13, Li), guidance=3.5)
    assert model.calibration_state()[0]
    ts = util_schedule(2, Li)

    def cmp(B, **knobs):
        with _lib.tuning(**knobs):
            alone = model.denoise(*sl(1), ts, guidance=3.5)
            whole = model.denoise(*sl(B), ts, guidance=3.5)
            torch.cuda.synchronize()
        same = torch.equal(alone[0].view(torch.int16), whole[0].view(torch.int16))
        rel = ((alone[0].float() - whole[0].float()).norm() / alone[0].float().norm()).item()
        nb = sum(torch.equal(model.denoise(*(inp[k][i:i + 1].contiguous() for k in keys), ts, guidance=3.5)[0].view(torch.int16), whole[i].view(torch.int16)) for i in range(1, min(B, 4)))
        print(f"B = {B:2d} {str(knobs):40s} sample 0 alone == in batch: {'bit-identical' if same else 'rel-L2 %.3e' % rel}; samples 1..{min(B, 4) - 1}: {nb} of {min(B, 4) - 1} identical", flush=True)
        return same

    bad = [B for B in (int(v) for v in a.B.split(",")) if not cmp(B)]
    if bad:
        B = bad[0]
        for knobs in (dict(qlut=0), dict(gemm_hybrid=0), dict(gemm_persist=0), dict(gemm_tile192=0), dict(w_pairs=0), dict(prefetch=0), dict(attn_split=0), dict(fuse_kv=1),
                      dict(gemm_esel=0), dict(gemm_hybrid=0, gemm_persist=0, qlut=0)):
            try:
                cmp(B, **knobs)
            except Exception as e:  # noqa: BLE001
                print(f"B = {B} {knobs}: {type(e).__name__}: {e}", flush=True)
