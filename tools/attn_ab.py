"""Interleaved A/B of the attention kernels in ONE process (guide rule 24): the 8-wave kernel (attention2.hip) with bf16 K (unfolded) and fp16 K
(folded: softmax scale in Q, running max in the accumulator init, f16 MFMAs for QK^T; the default) vs the 4-wave kernel (attention4.hip, fp16 K,
FLUXMI_ATTN_V=4); FLUXMI_ATTN_VAR=2 = exact running max, FLUXMI_ATTN_ABL 8 = 4-byte fp8 stores, 2 = no barrier [timing only].  Flux-dev shapes, random data.
    python tools/attn_ab.py [--rounds 5] [--iters 20] [--L 4608 2816]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import ops

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, nargs="+", default=[4608, 2816, 8192]); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
VARIANTS = [("8-wave bf16 K, 4 B stores", {"FLUXMI_ATTN_ABL": "8"}, False), ("8-wave bf16 K", {}, False),
            ("8-wave fp16 K (folded)", {}, True), ("8-wave folded, no barrier*", {"FLUXMI_ATTN_ABL": "2"}, True),
            ("8-wave folded, exact max", {"FLUXMI_ATTN_VAR": "2"}, True), ("4-wave fp16 K", {"FLUXMI_ATTN_V": "4"}, True),
            ("4-wave, exact max", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN_VAR": "2"}, True)]


def setenv(env):
    for k in ("FLUXMI_ATTN_V", "FLUXMI_ATTN_VAR", "FLUXMI_ATTN_ABL"):
        os.environ.pop(k, None)
    os.environ.update(env)


for L in a.L:
    B, H = 1, 24
    Lp = (L + 63) // 64 * 64
    q = torch.randn(B, H, L, 128, device=dev).bfloat16(); k = torch.randn(B, H, L, 128, device=dev).bfloat16()
    vt = torch.randn(B, H, 128, Lp, device=dev).bfloat16()
    one = torch.tensor(1.0, device=dev)
    o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
    k16 = k.half()
    res = {n: [] for n, _, _ in VARIANTS}
    for n, env, f16 in VARIANTS:
        setenv(env)
        for _ in range(3): ops.attention(q, k16 if f16 else k, vt, q_scale0=one, out=o8)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for n, env, f16 in VARIANTS:
            setenv(env)
            kk = k16 if f16 else k
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): ops.attention(q, kk, vt, q_scale0=one, out=o8)
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / a.iters * 1e-3)
    fl = 4 * L * L * 128 * H * B
    for n, _, _ in VARIANTS:
        ts = sorted(res[n]); t = ts[len(ts) // 2]
        print(f"L={L:5d} {n:24s}: median {fl / t / 1e12:7.1f} TF/s ({t * 1e6:6.1f} us)  best {fl / ts[0] / 1e12:7.1f}  frac of 2.5 PF {fl / t / 2.5e15:.3f}", flush=True)
