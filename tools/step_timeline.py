"""Per-tile timeline of the persistent GEMM INSIDE the denoise step (not in a back-to-back probe loop): the engine runs frozen steps
eagerly with the timing build of tile config 18 selected (fluxmi_tuning_t.gemm_persist = 2); every persistent launch overwrites the same
debug buffer, so after a step it holds the step's LAST persistent launch (the last single block's linear1).  Prints K-loop / epilogue
cycles per tile and the shader clock the workgroups saw -- the numbers tools/gemm_probe.py --timeline gives for the isolated kernel.
    python tools/step_timeline.py [--steps 4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
sys.path.insert(0, ROOT)
import torch

import util
from bench import util_schedule
from float8_quantize import quantize_flow_transformer_and_dispatch_float8
from fluxmi import _lib, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
    p = cfg.params
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=0, device=dev))
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=True, quantize_flow_embedder_layers=False)
        inp = synth.make_inputs(p, 1024, 1024, 512, batch=1, seed=0)
        d = {k: v.to(dev) for k, v in inp.items()}
        Li = d["img"].shape[1]
        model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], util_schedule(13, Li), guidance=3.5, use_graph=False)
        nwg = 256
        dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        _lib.call("fluxmi_gemm_debug_buffer", dbg.data_ptr())
        with _lib.tuning(gemm_persist=2):
            for graph in (False, True):
                model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], util_schedule(a.steps, Li), guidance=3.5, use_graph=graph)
                torch.cuda.synchronize()
                t = dbg.cpu().view(nwg, 8, 8)
                used = t[:, :, 0] != 0
                kl = (t[:, :, 1] - t[:, :, 0])[used].float()
                ep = (t[:, :, 2] - t[:, :, 1])[used].float()
                ntile = used.sum(1)
                wg = torch.nonzero(ntile >= 2).flatten()
                last = (ntile[wg] - 1).clamp(max=7)
                cyc = (t[wg, last, 2] - t[wg, 0, 2]).float()
                rt = (t[wg, last, 3] - t[wg, 0, 3]).float() * 10e-9
                clk = (cyc / rt)[rt > 0]
                span = (t[wg, last, 2] - t[wg, 0, 0]).float()
                print(f"in-step ({'hipGraph replay' if graph else 'eager'}), last persistent launch of the step: {int(used.sum())} tiles on {int(used.any(1).sum())} workgroups; "
                      f"K loop {kl.mean():.0f} cycles (min {kl.min():.0f} max {kl.max():.0f}), epilogue {ep.mean():.0f}; per-workgroup span {span.mean():.0f} cycles; "
                      f"shader clock {clk.median().item() / 1e9:.3f} GHz (min {clk.min().item() / 1e9:.3f} max {clk.max().item() / 1e9:.3f})", flush=True)
                for j in range(8):
                    u = used[:, j]
                    if u.any():
                        print(f"   tile #{j}: {int(u.sum())} workgroups, K loop {(t[:, j, 1] - t[:, j, 0])[u].float().mean():.0f}, epilogue {(t[:, j, 2] - t[:, j, 1])[u].float().mean():.0f}")
        _lib.call("fluxmi_gemm_debug_buffer", None)


if __name__ == "__main__":
    main()
