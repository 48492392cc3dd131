"""Attention kernel timing at the Flux-dev 1024^2 shape: python tools/attn_probe.py [--L 4608] [--iters 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import ops

ap = argparse.ArgumentParser(); ap.add_argument("--L", type=int, default=4608); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--bf16-k", action="store_true", help="bf16 K (unfolded kernel) instead of the engine's fp16 K (folded kernel)")
ap.add_argument("--split", default=None, help="comma list of fluxmi_tuning_t.attn_split values to time in turn, e.g. 0,2 (2 = balanced grid wherever a plan exists; default: the library's setting)")
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, H, L = a.B, 24, a.L
Lp = (L + 63) // 64 * 64
q = torch.randn(B, H, L, 128, device=dev).bfloat16(); k = torch.randn(B, H, L, 128, device=dev).bfloat16()
if not a.bf16_k:
    k = k.half()
vt = torch.randn(B, H, 128, Lp, device=dev).bfloat16()
one = torch.tensor(1.0, device=dev)
o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
from fluxmi import _lib
for sp in ([None] if a.split is None else [int(x) for x in a.split.split(",")] * 2):
    knobs = {} if sp is None else dict(attn_split=sp)
    with _lib.tuning(**knobs):
        for _ in range(3): ops.attention(q, k, vt, q_scale0=one, out=o8)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): ops.attention(q, k, vt, q_scale0=one, out=o8)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.iters * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    plan = ops.attention_plan(B, L, H) if (sp is None or sp) and not a.bf16_k else None
    if plan and (sp is None or sp == 1) and not plan["thin"]:
        plan = None  # attn_split = 1: thin last rounds only
    print(f"attention B={B} H={H} L={L} attn_split={'default' if sp is None else sp}"
          f" ({'balanced grid: %d whole tasks + %d pieces per XCD' % (plan['full_per_x'], len(plan['pieces'])) if plan else 'one workgroup per task'}): "
          f"{4 * L * L * 128 * H * B / t / 1e12:7.1f} TF/s ({t * 1e6:.1f} us)", flush=True)
