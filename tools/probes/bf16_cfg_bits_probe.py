"""Do the bf16 GEMM tile configs give the same bits?  One activation matrix, one weight, every config (and the automatic choice at two row counts)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch

from fluxmi import _lib, ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(16384, 3072, 64), (2048, 3072, 4096), (4096, 3072, 3072), (16384, 64, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    outs = {}
    for c in (2, 13, 16, 15, 100, -1):
        try:
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.linear(a, w, bias, out=o, tile_cfg=c)
            torch.cuda.synchronize()
            outs[c] = o
        except RuntimeError as e:
            print(f"M={M} N={N} K={K} cfg {c}: {str(e)[:100]}")
    # the automatic choice on the first quarter of the rows alone
    o = torch.empty(M // 4, N, dtype=torch.bfloat16, device=dev)
    ops.linear(a[:M // 4].contiguous(), w, bias, out=o, tile_cfg=-1)
    torch.cuda.synchronize()
    ref = outs.get(2, next(iter(outs.values())))
    for c, o2 in outs.items():
        d = (o2.view(torch.int16) != ref.view(torch.int16)).float().mean().item()
        print(f"M={M} N={N} K={K} cfg {c:4d} vs cfg 2: {'identical' if d == 0 else 'differs in %.4f of the elements' % d}")
    d = (o.view(torch.int16) != outs[-1][:M // 4].view(torch.int16)).float().mean().item()
    print(f"M={M} N={N} K={K} auto on M/4 rows vs auto on M rows: {'identical' if d == 0 else 'differs in %.4f of the elements' % d}", flush=True)
