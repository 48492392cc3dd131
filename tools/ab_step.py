"""In-step A/B of kernel-selection knobs: ONE engine (Flux-dev 1024^2 by default, calibrated once), the variants run round-robin in the
same process on the same box, each as hipGraph-replayed requests bracketed by synchronize -- the only comparison that means anything on
this pool (boxes differ by 5 %, and an isolated probe loop flatters a kernel: DESIGN.md section 5).
    python tools/ab_step.py --variant base: --variant nosplit:attn_split=0 [--variant name:knob=v,knob=v ...] [--steps 20] [--rounds 3]
                            [--height 1024 --width 1024] [--batch 1] [--check]
Prints ms/step per variant and round, the median, the sustained shader clock, and (--check) whether the latents of every variant are
bit-identical to the first variant's."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
sys.path.insert(0, ROOT)
import torch

import util
from bench import _clock_from_samples, util_schedule
from float8_quantize import quantize_flow_transformer_and_dispatch_float8
from fluxmi import _lib, ops, synth


def parse_variant(s):
    name, _, rest = s.partition(":")
    knobs = {}
    for kv in filter(None, rest.split(",")):
        k, v = kv.split("=")
        knobs[k] = float(v) if k == "attn_defer_log2" else int(v)
    return name, knobs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--embedders", action="store_true", help="quantize_flow_embedder_layers (BASELINE config 3)")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    variants = [parse_variant(v) for v in (a.variant or ["base:"])]
    dev = torch.device("cuda:0")
    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=a.embedders)
    p = cfg.params
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=0, device=dev))
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=True, quantize_flow_embedder_layers=a.embedders)
        inp = synth.make_inputs(p, a.height, a.width, 512, batch=a.batch, seed=0)
        d = {k: v.to(dev) for k, v in inp.items()}
        Li = d["img"].shape[1]
        run = lambda n, graph=True: model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], util_schedule(n, Li), guidance=3.5, use_graph=graph)
        run(13, False)  # calibration
        torch.cuda.synchronize()
        res = {n: [] for n, _ in variants}
        clk = {n: [] for n, _ in variants}
        lat = {}
        samples = torch.zeros(2 * 8 * 3, dtype=torch.int64, device=dev)
        for r in range(a.rounds):
            for name, knobs in variants:
                with _lib.tuning(**knobs):
                    out = run(3)  # capture under these knobs + warm
                    torch.cuda.synchronize()
                    _lib.call("fluxmi_clock_sample", samples.data_ptr(), ops._stream())
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = run(a.steps)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t0
                    _lib.call("fluxmi_clock_sample", samples.data_ptr() + 8 * 3 * 8, ops._stream())
                    torch.cuda.synchronize()
                res[name].append(el / a.steps * 1e3)
                clk[name].append(_clock_from_samples(samples.cpu().view(2, 8, 3))[0])
                if r == 0:
                    lat[name] = out.clone()
        base = variants[0][0]
        for name, knobs in variants:
            v = sorted(res[name])
            med = v[len(v) // 2]
            rel = med / sorted(res[base])[len(res[base]) // 2] - 1
            same = ""
            if a.check and name != base:
                same = "  latents == first variant: " + ("bit-identical" if torch.equal(lat[name].view(torch.int16), lat[base].view(torch.int16)) else
                                                         f"rel-L2 {((lat[name].float() - lat[base].float()).norm() / lat[base].float().norm()).item():.3e}")
            print(f"{name:24s} {knobs}  ms/step {['%.3f' % x for x in res[name]]}  median {med:.3f} ({rel:+.2%} vs {base})  clock GHz {clk[name]}{same}", flush=True)


if __name__ == "__main__":
    main()
