"""Interleaved A/B of the two LayerNorm + modulate + quantise kernels in ONE process: FLUXMI_LN_V=1 (one wave per row, every row resident
at once) vs FLUXMI_LN_V=2 (streaming: one 8-wave workgroup per CU, next row's loads under this row's arithmetic, paired bf16 roundings).
    python tools/ln_ab.py [--rounds 5] [--iters 50]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import ops

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, nargs="+", default=[4608, 2816]); ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
H = 3072
for L in a.L:
    Lt = 512
    x = (torch.randn(1, L, H, device=dev) * 2).bfloat16()
    mods = (torch.randn(1, 4 * H, device=dev) * 0.5).bfloat16()
    v = lambda i: mods[:, i * H:(i + 1) * H]
    q0, q1 = torch.tensor(900.0, device=dev), torch.tensor(2000.0, device=dev)
    run = lambda: ops.ln_modulate(x, v(0), v(1), v(2), v(3), split=Lt, q_scale0=q0, q_scale1=q1)
    outs, res = {}, {"1": [], "2": []}
    for var in ("1", "2"):
        os.environ["FLUXMI_LN_V"] = var
        for _ in range(3): outs[var] = run()
    torch.cuda.synchronize()
    same = (outs["1"].view(torch.uint8) == outs["2"].view(torch.uint8)).float().mean().item()
    for r in range(a.rounds):
        for var in ("1", "2"):
            os.environ["FLUXMI_LN_V"] = var
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): run()
            e1.record(); torch.cuda.synchronize()
            res[var].append(e0.elapsed_time(e1) / a.iters * 1e-3)
    by = L * H * 3
    for var in ("1", "2"):
        ts = sorted(res[var]); t = ts[len(ts) // 2]
        print(f"L={L:5d} FLUXMI_LN_V={var}: median {t * 1e6:6.2f} us  {by / t / 1e12:5.2f} TB/s algorithmic (best {ts[0] * 1e6:6.2f} us)   identical bytes v1==v2: {same:.6f}", flush=True)
