"""Isolated A/B of the two attention kernels in the ENGINE's call form (raw-Q mode: QKNorm + RoPE applied while the Q fragments are built; fp16 K;
fp8 output) at the Flux-dev shape: python tools/attn_rawq_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, H, L, Lt = 1, 24, 4608, 512
qkv = torch.randn(B, L, 3 * H * 128, device=dev).bfloat16()
pe = torch.randn(B, L, 64, 2, device=dev).bfloat16()
s = [(1 + 0.1 * torch.randn(128, device=dev)).bfloat16() for _ in range(4)]
_, K16, VT = ops.qkv_rope(qkv, pe, s[0], s[1], s[2], s[3], split=Lt, heads=H, skip_q=True, k_f16=True)
one = torch.tensor(1.0, device=dev)
o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
Q, _, _ = ops.qkv_rope(qkv, pe, s[0], s[1], s[2], s[3], split=Lt, heads=H, k_f16=True)
Qr = torch.randn(B, H, L, 128, device=dev).bfloat16(); Kr = torch.randn(B, H, L, 128, device=dev).half(); VTr = torch.randn(B, H, 128, L, device=dev).bfloat16()
variants = [("8-wave raw-Q", None, True), ("4-wave raw-Q", "4", True), ("8-wave Q tensor", None, False), ("4-wave Q tensor", "4", False),
            ("8-wave randn Q/K/V", None, "r"), ("4-wave randn Q/K/V", "4", "r"), ("8-wave randn V only", None, "v"), ("4-wave randn V only", "4", "v"),
            ("8-wave randn Q,K only", None, "qk"), ("4-wave randn Q,K only", "4", "qk")]
def run(v, raw):
    os.environ.pop("FLUXMI_ATTN_V", None)
    if v: os.environ["FLUXMI_ATTN_V"] = v
    if raw is True: ops.attention_rawq(qkv, pe, s[0], K16, VT, qn_scale1=s[2], split=Lt, q_scale0=one, q_scale1=one, out=o8)
    elif raw == "r": ops.attention(Qr, Kr, VTr, q_scale0=one, out=o8)
    elif raw == "v": ops.attention(Q, K16, VTr, q_scale0=one, out=o8)
    elif raw == "qk": ops.attention(Qr, Kr, VT, q_scale0=one, out=o8)
    else: ops.attention(Q, K16, VT, q_scale0=one, out=o8)
res = {n: [] for n, _, _ in variants}
for n, v, raw in variants:
    for _ in range(3): run(v, raw)
torch.cuda.synchronize()
for r in range(5):
    for n, v, raw in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(v, raw)
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 20 * 1e3)
print("Q rms", Q.float().pow(2).mean().sqrt().item(), "K rms", K16.float().pow(2).mean().sqrt().item(), "VT rms", VT.float().pow(2).mean().sqrt().item())
for n, _, _ in variants:
    ts = sorted(res[n]); print(f"{n:22s}: median {ts[2]:7.1f} us  best {ts[0]:7.1f}", flush=True)
