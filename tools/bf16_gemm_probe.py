"""bf16 small-M GEMM probe: one shape, several tile configs / split factors interleaved (113 + S = split-K with S splits).
    python tools/bf16_gemm_probe.py --shape 512,21504,3072 --ab 2,13,16,115,116"""
import argparse, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import _lib, ops
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="512,21504,3072"); ap.add_argument("--ab", default="2,13,16"); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a_ = ap.parse_args()
M, N, K = (int(v) for v in a_.shape.split(","))
dev = torch.device("cuda:0"); torch.manual_seed(0)
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
cfgs = [int(c) for c in a_.ab.split(",")]
ok = []
for c in cfgs:
    try:
        ops.linear(a, w, bias, out=out, tile_cfg=c); torch.cuda.synchronize(); ok.append(c)
    except RuntimeError as e:
        print(f"cfg {c}: {e}")
res = {c: [] for c in ok}
for _ in range(a_.rounds):
    for c in ok:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a_.iters): ops.linear(a, w, bias, out=out, tile_cfg=c)
        e1.record(); torch.cuda.synchronize()
        res[c].append(e0.elapsed_time(e1) / a_.iters * 1e3)
for c in ok:
    t = statistics.median(res[c])
    print(f"bf16 M={M} N={N} K={K} cfg={c}: median {t:7.1f} us  {2*M*N*K/t/1e6:7.1f} TF/s  weights {N*K*2/t/1e3:7.1f} GB/s")
