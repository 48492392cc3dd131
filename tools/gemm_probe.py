"""One GEMM shape / tile config / epilogue, launched `iters` times -- the target for rocprofv3 PMC passes and A/B timing:
    python tools/gemm_probe.py --shape 4608,21504,3072 --cfg 13 --epi bf16 --iters 10 [--fill zero] [--ab 13,16]
With --ab a,b the two configs are timed interleaved over several rounds (median / min per config)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch

from fluxmi import _lib, ops

SHAPES = {
    "qkv": (4096, 9216, 3072), "proj": (4096, 3072, 3072), "mlp0": (4096, 12288, 3072), "mlp2": (4096, 3072, 12288),
    "lin1": (4608, 21504, 3072), "lin2": (4608, 3072, 15360), "tqkv": (512, 9216, 3072), "tmlp2": (512, 3072, 12288),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="lin1")
    ap.add_argument("--cfg", type=int, default=13)
    ap.add_argument("--epi", default="bf16", choices=["bf16", "gelu", "gate", "quant"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fill", default="rand", choices=["rand", "zero"])
    ap.add_argument("--ab", default=None, help="comma list of configs to A/B interleaved")
    ap.add_argument("--rounds", type=int, default=7)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M, N, K = SHAPES[args.shape] if args.shape in SHAPES else tuple(int(v) for v in args.shape.split(","))
    torch.manual_seed(0)
    one = torch.tensor(1.0, device=dev)
    if args.fill == "zero":
        a = torch.zeros(M, K, device=dev).to(torch.float8_e5m2)
        w = torch.zeros(N, K, device=dev).to(torch.float8_e4m3fn)
    else:
        a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2)
        w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
    bias = torch.randn(N, device=dev).bfloat16()
    gate = torch.randn(N, device=dev).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out8 = torch.empty(M, N, dtype=torch.float8_e5m2, device=dev)
    resid = torch.randn(M, N, device=dev).bfloat16()

    def run(cfg):
        if args.epi == "bf16":
            ops.linear(a, w, bias, one, one, out=out, tile_cfg=cfg)
        elif args.epi == "quant":
            ops.linear(a, w, bias, one, one, out=out8, epilogue=_lib.EPI_QUANT, q_scale=one, tile_cfg=cfg)
        elif args.epi == "gelu":
            ops.linear(a, w, bias, one, one, out=out8, epilogue=_lib.EPI_GELU_QUANT, q_scale=one, tile_cfg=cfg)
        else:
            ops.linear(a, w, bias, one, one, out=resid, resid=resid, gate=gate, epilogue=_lib.EPI_GATE_RESID, tile_cfg=cfg)

    def timed(cfg, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run(cfg)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    cfgs = [int(c) for c in args.ab.split(",")] if args.ab else [args.cfg]
    for c in cfgs:
        run(c)
    torch.cuda.synchronize()
    res = {c: [] for c in cfgs}
    for _ in range(args.rounds if args.ab else 1):
        for c in cfgs:
            res[c].append(timed(c, args.iters))
    for c in cfgs:
        tf = [2 * M * N * K / t / 1e12 for t in res[c]]
        print(f"{args.shape} M={M} N={N} K={K} epi={args.epi} fill={args.fill} cfg={c}: median {statistics.median(tf):7.1f} TF/s  max {max(tf):7.1f}  "
              f"({statistics.median(res[c]) * 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
