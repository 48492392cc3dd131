#!/usr/bin/env python
"""CPU baseline of bench.py: the reference's bf16 flow path (nn.Linear, no fp8 -- BASELINE.md section 3) timed on the host cores.
TEST / BENCH INFRASTRUCTURE (never on the product path).  Prints one JSON object.

kind "reference": the UNMODIFIED /root/reference modules (through oracle/ref_shims.py) when that tree exists (build container);
kind "port": oracle/flux_oracle.py, which oracle/gen_golden*.py pin bit-for-bit to the reference (the GPU box has no /root/reference).
--full-step (what bench.py asks for): the whole 19+38-block step executed in full, one block's weights reused for every block of its kind
(synthesising 24 GB of distinct random weights would take minutes; the arithmetic and the streamed bytes are the same: a block's
0.6 GB of bf16 weights do not fit any host cache) -- provided one DoubleStreamBlock + one SingleStreamBlock predict a step of at most
--full-budget seconds (75); otherwise, and without --full-step: those two blocks timed and extrapolated x19 / x38.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd", "fluxmi"))  # synth.py as a plain module (no package import, no libfluxmi)

import warnings

warnings.filterwarnings("ignore")
import torch

import flux_oracle as fo
import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--txt-len", type=int, default=512)
    ap.add_argument("--schnell", action="store_true")
    ap.add_argument("--full-step", action="store_true")
    ap.add_argument("--budget", type=float, default=25.0)
    ap.add_argument("--full-budget", type=float, default=75.0)
    ap.add_argument("--port", action="store_true", help="force the oracle port even when /root/reference exists")
    a = ap.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    p = fo.FluxParams(depth=1, depth_single_blocks=1, guidance_embed=not a.schnell)
    sd = synth.make_state_dict(p, seed=0)
    Li, Lt, H = (a.height // 16) * (a.width // 16), a.txt_len, p.hidden_size
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, Li, H, generator=g).bfloat16()
    txt = torch.randn(1, Lt, H, generator=g).bfloat16()
    vec = torch.randn(1, H, generator=g).bfloat16()
    img_ids, txt_ids = fo.make_ids(1, a.height // 16, a.width // 16, Lt, torch.bfloat16)
    kind, dbl, sgl = "port", None, None
    if os.path.isdir("/root/reference") and not a.port:
        import ref_shims

        f8q, fm, rutil = ref_shims.import_reference()
        rcfg = rutil.load_config(rutil.ModelVersion.flux_schnell if a.schnell else rutil.ModelVersion.flux_dev, flow_dtype="bfloat16")
        rcfg.params.depth, rcfg.params.depth_single_blocks = 1, 1
        with torch.device("meta"):
            rm = fm.Flux(rcfg, dtype=torch.bfloat16)
            rm.type(torch.bfloat16)
        rm.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True, assign=True)
        rm.eval()
        pe = rm.pe_embedder(torch.cat((txt_ids, img_ids), 1))
        dbl = lambda: rm.double_blocks[0](img, txt, vec, pe)
        sgl = lambda: rm.single_blocks[0](torch.cat((txt, img), 1), vec, pe)
        kind = "reference"
    else:
        orc = fo.FluxOracle(sd, p, quantize=None)
        pe = fo.rope_table(torch.cat((txt_ids, img_ids), 1), p.axes_dim, p.theta, torch.bfloat16)
        dbl = lambda: orc.double_block(0, img, txt, vec, pe)
        sgl = lambda: orc.single_block(0, torch.cat((txt, img), 1), vec, pe)
    # thread count: the one that runs the real thing (one SingleStreamBlock call) fastest on this host -- containers often expose more
    # CPUs than their quota, and small-GEMM probes do not scale like the block does
    timing = {}
    with torch.inference_mode():
        for n in sorted({n for n in (avail, 64, 32, 16, 8, 4) if n <= avail} or {1}, reverse=True):
            torch.set_num_threads(n)
            sgl()
            t0 = time.time()
            sgl()
            timing[n] = time.time() - t0
    cores = min(timing, key=timing.get)
    torch.set_num_threads(cores)
    what = "the unmodified reference modules.flux_model (bf16 nn.Linear)" if kind == "reference" else "oracle port of the reference's bf16 flow path"
    with torch.inference_mode():
        t0 = time.time()
        dbl(); sgl()
        warm = time.time() - t0
        if a.full_step and 0.5 * (19 + 38) * warm <= a.full_budget:
            t0 = time.time()
            for _ in range(19):
                dbl()
            for _ in range(38):
                sgl()
            step_s = time.time() - t0
            sample = (f"{what}: the full step at {a.height}x{a.width} (L = {Li}+{Lt}), 19 DoubleStreamBlock + 38 SingleStreamBlock calls executed "
                      f"back to back = {step_s:.2f} s (one block's weights per kind; embedders / final layer < 1 % not included)")
        else:
            reps = max(1, min(5, int(a.budget / max(warm, 1e-3)) - 1))
            t0 = time.time()
            for _ in range(reps):
                dbl()
            td_ = (time.time() - t0) / reps
            t0 = time.time()
            for _ in range(reps):
                sgl()
            ts_ = (time.time() - t0) / reps
            step_s = 19 * td_ + 38 * ts_
            sample = (f"{what}: 1 DoubleStreamBlock ({td_:.3f} s) + 1 SingleStreamBlock ({ts_:.3f} s) at L={Li + Lt}, {reps} reps each, "
                      f"extrapolated to 19+38 blocks = {step_s:.1f} s/step")
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    # `cores` = the threads the timed run used (the contract's field); host_cores = logical CPUs of the box, host_cores_available = the ones
    # this process may run on (affinity mask / container quota), thread_scan = seconds of one SingleStreamBlock call per candidate count
    print(json.dumps({"value": 1.0 / step_s, "unit": "it/s", "cores": cores, "threads_used": cores, "host_cores": os.cpu_count(),
                      "host_cores_available": avail, "cpu_model": model, "thread_scan_s": {str(k): round(v, 3) for k, v in sorted(timing.items())},
                      "kind": kind, "sample": sample}))


if __name__ == "__main__":
    main()
