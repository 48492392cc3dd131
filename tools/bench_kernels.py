"""Kernel micro-benchmarks on the Flux-dev shapes (run on the GPU box):
    python tools/bench_kernels.py [--out gpurun_out/kernels.txt]
GEMM: every tile config on every F8Linear shape of SURVEY.md Appendix C (TF/s); attention (TF/s);
row kernels (GB/s).  Timed with HIP events on the launch stream, random data (never zero-filled)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch

from fluxmi import _lib, ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    torch.manual_seed(0)
    one = torch.tensor(1.0, device=dev)
    shapes = [
        ("double.img qkv", 4096, 9216, 3072), ("double.img proj", 4096, 3072, 3072), ("double.img mlp0", 4096, 12288, 3072),
        ("double.img mlp2", 4096, 3072, 12288), ("double.txt qkv", 512, 9216, 3072), ("double.txt mlp2", 512, 3072, 12288),
        ("single lin1", 4608, 21504, 3072), ("single lin2", 4608, 3072, 15360),
    ]
    if args.quick:
        shapes = shapes[:2]
    CFGS = (13, 16, 2)
    emit("# fp8 (e5m2 x e4m3) GEMM, TF/s per tile config [13: 256x256 ping-pong ring | 16: 256x256 one wave per SIMD | 2: 128x128 double-buffered]")
    for name, M, N, K in shapes:
        a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2)
        w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
        bias = torch.randn(N, device=dev).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        row = []
        for cfg in CFGS:
            try:
                t = timeit(lambda: ops.linear(a, w, bias, one, one, out=out, tile_cfg=cfg))
                row.append(f"{2 * M * N * K / t / 1e12:7.1f}")
            except RuntimeError as ex:
                row.append("   n/a ")
        emit(f"{name:18s} M={M:5d} N={N:5d} K={K:5d}  " + " ".join(row))
    emit("# fused epilogues, 256x256 ping-pong ring (TF/s): bf16 | gelu+quant | gate+resid")
    for name, M, N, K in shapes[:4]:
        a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2)
        w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
        bias = torch.randn(N, device=dev).bfloat16()
        gate = torch.randn(N, device=dev).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        out8 = torch.empty(M, N, dtype=torch.float8_e5m2, device=dev)
        resid = torch.randn(M, N, device=dev).bfloat16()
        r = []
        r.append(timeit(lambda: ops.linear(a, w, bias, one, one, out=out, tile_cfg=13)))
        r.append(timeit(lambda: ops.linear(a, w, bias, one, one, out=out8, epilogue=_lib.EPI_GELU_QUANT, q_scale=one, tile_cfg=13)))
        r.append(timeit(lambda: ops.linear(a, w, bias, one, one, out=resid, resid=resid, gate=gate, epilogue=_lib.EPI_GATE_RESID, tile_cfg=13)))
        emit(f"{name:18s} " + " ".join(f"{2 * M * N * K / t / 1e12:7.1f}" for t in r))
    emit("# bf16 GEMM 4096x3072x3072 (TF/s) per cfg")
    a = torch.randn(4096, 3072, device=dev).bfloat16()
    w = torch.randn(3072, 3072, device=dev).bfloat16()
    out = torch.empty(4096, 3072, dtype=torch.bfloat16, device=dev)
    emit(" ".join(f"{2 * 4096 * 3072 * 3072 / timeit(lambda: ops.linear(a, w, out=out, tile_cfg=c)) / 1e12:7.1f}" for c in CFGS))

    emit("# attention fwd B=1 H=24 L=4608 D=128 (TF/s, 4*L^2*D*H flop)")
    B, H, L = 1, 24, 4608
    q = torch.randn(B, H, L, 128, device=dev).bfloat16()
    k = torch.randn(B, H, L, 128, device=dev).bfloat16()
    vt = torch.randn(B, H, 128, L, device=dev).bfloat16()
    o = torch.empty(B, L, H * 128, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.attention(q, k, vt, out=o))
    emit(f"bf16 out: {4 * L * L * 128 * H / t / 1e12:7.1f} TF/s  ({t * 1e6:.0f} us)")
    o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
    t = timeit(lambda: ops.attention(q, k, vt, q_scale0=one, out=o8))
    emit(f"fp8 out : {4 * L * L * 128 * H / t / 1e12:7.1f} TF/s  ({t * 1e6:.0f} us)")

    emit("# row kernels (GB/s of algorithmic traffic)")
    x = torch.randn(1, L, 3072, device=dev).bfloat16()
    mods = torch.randn(1, 6 * 3072, device=dev).bfloat16()
    t = timeit(lambda: ops.ln_modulate(x, mods[:, :3072], mods[:, 3072:6144], q_scale0=one))
    emit(f"ln_modulate->fp8 [4608,3072]: {L * 3072 * 3 / t / 1e9:7.0f} GB/s ({t * 1e6:.1f} us)")
    qkv = torch.randn(1, L, 9216, device=dev).bfloat16()
    pe = torch.randn(1, L, 64, 2, device=dev).bfloat16()
    s = torch.ones(128, device=dev).bfloat16()
    t = timeit(lambda: ops.qkv_rope(qkv, pe, s, s, heads=24))
    emit(f"qkv_rope [4608,9216]: {L * 9216 * 4 / t / 1e9:7.0f} GB/s ({t * 1e6:.1f} us)")
    # batched modulation GEMV: 19*2*18432 + 38*9216 rows of K=3072 fp8 ~ 3.2 GB
    Wm = (torch.randn(18432 * 8, 3072, device=dev)).to(torch.float8_e4m3fn)
    v = torch.randn(1, 3072, device=dev).bfloat16()
    t = timeit(lambda: ops.gemv(v, Wm, None, one, one, one), iters=5)
    emit(f"gemv fp8 N={Wm.shape[0]} K=3072 B=1: {Wm.numel() / t / 1e9:7.0f} GB/s ({t * 1e6:.0f} us)")
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
