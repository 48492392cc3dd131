"""Timing-only ablations of attention4 (FLUXMI_ATTN4_ABL): where does a lone wave's time go?
Needs a library built with the ablation variants:  make -C flux-fp8-api_amd/csrc EXTRA=-DFLUXMI_ATTN4_ABLATIONS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import ops
dev = torch.device("cuda:0")
L, B, H = 4608, 1, 24
Hs = [int(x) for x in sys.argv[1:]] or [24]
for H in Hs:
    q = torch.randn(B, H, L, 128, device=dev).bfloat16(); k16 = torch.randn(B, H, L, 128, device=dev).half()
    vt = torch.randn(B, H, 128, L, device=dev).bfloat16(); one = torch.tensor(1.0, device=dev)
    o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
    variants = [("v2 (8x32)", {"FLUXMI_ATTN_V": "2"}), ("v4", {"FLUXMI_ATTN_V": "4"}), ("v4 -softmax VALU", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "1"}), ("v4 -DMA", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "2"}),
                ("v4 -barrier/vmcnt", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "4"}), ("v4 -ds_read", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "8"}), ("v4 -finish/decision", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "16"}),
                ("v4 -MFMA", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "32"}), ("v4 MFMA+barrier only", {"FLUXMI_ATTN_V": "4", "FLUXMI_ATTN4_ABL": "27"})]
    def setenv(env):
        for kk in ("FLUXMI_ATTN_V", "FLUXMI_ATTN4_ABL"): os.environ.pop(kk, None)
        os.environ.update(env)
    res = {n: [] for n, _ in variants}
    for n, env in variants:
        setenv(env)
        for _ in range(3): ops.attention(q, k16, vt, q_scale0=one, out=o8)
    torch.cuda.synchronize()
    for r in range(5):
        for n, env in variants:
            setenv(env)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.attention(q, k16, vt, q_scale0=one, out=o8)
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 10 * 1e-3)
    fl = 4 * L * L * 128 * H * B
    for n, _ in variants:
        ts = sorted(res[n]); t = ts[len(ts) // 2]
        print(f"H={H:3d} ({(L // 256) * H} WGs) {n:22s}: {t * 1e6:7.1f} us  ({fl / t / 1e12:7.1f} TF/s equiv)", flush=True)
