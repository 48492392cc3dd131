"""Does an independent GEMM hide in the tail of the attention grid?  (432 workgroups on 256 CUs = 1.69 rounds.)
Times attention (stream A) and the mlp part of SingleStreamBlock.linear2 -- M=4608, N=3072, K=12288, which does not depend on
attention -- back to back on one stream vs concurrently on two streams, each pattern captured in a hipGraph and replayed.
    python tools/overlap_probe.py [--iters 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import _lib, ops

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--K2", type=int, default=12288)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, L = 24, 4608
q = torch.randn(1, H, L, 128, device=dev).bfloat16(); k = torch.randn(1, H, L, 128, device=dev).bfloat16()
vt = torch.randn(1, H, 128, L, device=dev).bfloat16()
one = torch.tensor(1.0, device=dev)
o8 = torch.empty(1, L, H * 128, dtype=torch.float8_e5m2, device=dev)
M, N, K = L, 3072, a.K2
A = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2); W = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
bias = torch.randn(N, device=dev).bfloat16(); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
A3 = (torch.randn(M, 3072, device=dev) * 2).to(torch.float8_e5m2); W3 = (torch.randn(N, 3072, device=dev) * 0.5).to(torch.float8_e4m3fn)
out3 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

def attn(): ops.attention(q, k, vt, q_scale0=one, out=o8)
def gemm(): ops.linear(A, W, bias, one, one, out=out)
def gemm3(): ops.linear(A3, W3, bias, one, one, out=out3)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def seq():
    attn(); gemm(); gemm3()

def par():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        gemm()
    attn()
    cur.wait_stream(s2)
    gemm3()

def timed(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        for _ in range(a.iters): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.iters * 1e3)
    print(f"{name}: {sorted(ts)[2]:8.1f} us per (attention + gemm K={K} + gemm K=3072)", flush=True)

def single(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / a.iters * 1e3:8.1f} us", flush=True)

single(attn, "attention alone"); single(gemm, f"gemm K={K} alone"); single(gemm3, "gemm K=3072 alone")
timed(seq, "sequential (one stream) ")
timed(par, "concurrent (two streams)")
timed(seq, "sequential (one stream) ")
timed(par, "concurrent (two streams)")
