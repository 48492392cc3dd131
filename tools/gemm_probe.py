"""One GEMM shape / tile config / epilogue, launched `iters` times -- the target for rocprofv3 PMC passes and A/B timing:
    python tools/gemm_probe.py --shape 4608,21504,3072 --cfg 13 --epi bf16 --iters 10 [--fill zero] [--ab 13,18]
With --ab a,b the configs are timed interleaved over several rounds (median / best per config) and, with --check, their outputs are
compared byte for byte first (the persistent kernel 18 runs the K loop of 13 in the same order: identical bits are the expectation).
    --epi split       single-block linear1: q|k|v columns as bf16 (V^T fused with --vt), gelu(mlp) columns as fp8 through the table
    --groups 512,4096 a grouped launch (double-block txt + img streams, one weight matrix each)
    --timeline        tile config 19 (= 18 with in-kernel timestamps): per-tile K-loop / epilogue durations and the shader clock"""
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
H = 3072


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="lin1")
    ap.add_argument("--cfg", type=int, default=13)
    ap.add_argument("--epi", default="bf16", choices=["bf16", "gelu", "gate", "quant", "split"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fill", default="rand", choices=["rand", "zero"])
    ap.add_argument("--ab", default=None, help="comma list of configs to A/B interleaved")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--groups", default=None, help="comma list of M per group (grouped launch, one weight matrix per group)")
    ap.add_argument("--vt", action="store_true", help="bf16 / split epilogue: fused V^T output for the columns [2H, 3H)")
    ap.add_argument("--no-lut", action="store_true", help="quantising epilogues without the table (VALU GELU)")
    ap.add_argument("--check", action="store_true", help="with --ab: compare the outputs of the configs byte for byte")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--pairs", action="store_true", help="give every group a row-pair copy of its weight (fluxmi_gemm_group_t.W_pairs, tile config 18 / 19)")
    ap.add_argument("--touch", default="none", choices=["none", "w", "a", "wa"], help="read these operands of the NEXT launch with a plain streaming kernel right "
                    "before it (emulates a weight prefetch into the memory-side cache; with --rotate)")
    ap.add_argument("--rotate-what", default="all", choices=["all", "w", "a"], help="which operands differ between the rotated sets (the others are shared)")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many copies of the operands, one per launch: with enough of them (> 256 MB "
                    "in total) every launch finds its weights in HBM, not in the 256 MB Infinity Cache -- the situation inside the denoise step")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M, N, K = SHAPES[args.shape] if args.shape in SHAPES else tuple(int(v) for v in args.shape.split(","))
    Ms = [int(v) for v in args.groups.split(",")] if args.groups else [M]
    torch.manual_seed(0)
    one = torch.tensor(1.0, device=dev)
    qs = torch.tensor(3.0, device=dev)
    lut = None if args.no_lut else ops.build_quant_lut(qs, _lib.E5M2, act=1)
    epi = {"bf16": _lib.EPI_BF16, "gelu": _lib.EPI_GELU_QUANT, "gate": _lib.EPI_GATE_RESID, "quant": _lib.EPI_QUANT, "split": _lib.EPI_SPLIT}[args.epi]

    shared = {}

    def build():
        groups, keep, outs, aw = [], [], [], []
        for gi, Mg in enumerate(Ms):
            if args.fill == "zero":
                a = torch.zeros(Mg, K, device=dev).to(torch.float8_e5m2)
                w = torch.zeros(N, K, device=dev).to(torch.float8_e4m3fn)
            else:
                a = (torch.randn(Mg, K, device=dev) * 2).to(torch.float8_e5m2)
                w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
            if args.rotate_what == "w":
                a = shared.setdefault(("a", gi), a)
            if args.rotate_what == "a":
                w = shared.setdefault(("w", gi), w)
            bias = torch.randn(N, device=dev).bfloat16()
            kw = {}
            if args.epi == "bf16":
                o = torch.zeros(Mg, N, dtype=torch.bfloat16, device=dev)
            elif args.epi in ("quant", "gelu"):
                o = torch.zeros(Mg, N, dtype=torch.float8_e5m2, device=dev)
                kw = dict(q_scale=qs.data_ptr(), q_lut=lut.data_ptr() if (lut is not None and args.epi == "gelu") else None)
            elif args.epi == "gate":
                o = torch.randn(Mg, N, device=dev).bfloat16()
                gate = torch.randn(N, device=dev).bfloat16()
                keep.append(gate)
                kw = dict(gate=gate.data_ptr(), resid=o.data_ptr(), ldr=N)
            else:  # split: [q | k | v | mlp]
                o = torch.zeros(Mg, 3 * H, dtype=torch.bfloat16, device=dev)
                o2 = torch.zeros(Mg, 5 * H, dtype=torch.float8_e5m2, device=dev)
                outs.append(o2)
                kw = dict(C2=o2.data_ptr(), ldc2=5 * H, split_n=3 * H, c2_col0=H, q_scale=qs.data_ptr(), q_lut=lut.data_ptr() if lut is not None else None)
            if args.vt and args.epi in ("bf16", "split"):
                Lp = (Mg + 63) // 64 * 64
                vt = torch.zeros(H, Lp, dtype=torch.bfloat16, device=dev)
                outs.append(vt)
                kw.update(vt_out=vt.data_ptr(), vt_ld=Lp, tok0=0, vt_rows=Lp, kv_col0=H, heads=H // 128)
                if _lib.get_tuning().fuse_kv >= 2:  # K (QKNorm + RoPE, head-major) from the epilogue as well, as the engine issues it
                    kt = torch.zeros(H // 128, Mg, 128, dtype=torch.float16, device=dev)
                    pe = torch.randn(Mg, 64, 2, device=dev).bfloat16()
                    kn = (1 + 0.1 * torch.randn(128, device=dev)).bfloat16()
                    keep += [pe, kn]
                    outs.append(kt)
                    kw.update(k_out=kt.data_ptr(), pe=pe.data_ptr(), k_norm=kn.data_ptr(), k_rows=Mg, k_f16=True)
            if args.pairs:
                wp = ops.pair_rows(w)
                keep.append(wp)
                kw.update(W_pairs=wp.data_ptr())
            keep += [a, w, bias]
            aw.append((a, w))
            outs.append(o)
            groups.append(ops.make_group(a.data_ptr(), w.data_ptr(), bias.data_ptr(), one.data_ptr(), one.data_ptr(), o.data_ptr(), Mg, K, o.stride(0), **kw))
        return groups, keep, outs, aw

    sets = [build() for _ in range(max(1, args.rotate))]
    groups, keep, outs, _aw = sets[0]
    resid0 = [o.clone() for o in outs] if args.epi == "gate" else None
    turn = [0]

    def run(cfg):
        g, _k, _o, aw_ = sets[turn[0] % len(sets)]
        turn[0] += 1
        if args.touch != "none":
            for a_, w_ in aw_:
                if "w" in args.touch:
                    w_.view(torch.int32).max()
                if "a" in args.touch:
                    a_.view(torch.int32).max()
        ops.gemm_grouped(g, N, K, True, _lib.E5M2, epi, cfg)

    def reset():
        if resid0 is not None:  # gate*y + x updates the residual stream in place
            for o, r in zip(outs, resid0):
                o.copy_(r)

    def timed(cfg, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run(cfg)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    cfgs = [int(c) for c in args.ab.split(",")] if args.ab else [args.cfg]
    tag = f"{args.shape} Ms={Ms} N={N} K={K} epi={args.epi}{'+vt' if args.vt else ''}{' no-lut' if args.no_lut else ''} fill={args.fill}{f' rotate={len(sets)}({args.rotate_what})' if len(sets) > 1 else ''}{f' touch={args.touch}' if args.touch != 'none' else ''}"
    if args.check:
        turn[0] = 0
        assert len(sets) == 1, "--check compares the outputs of one operand set: do not combine with --rotate"
        ref = None
        for c in cfgs:
            reset()
            for o in outs:
                if resid0 is None:
                    o.zero_()
            run(c)
            torch.cuda.synchronize()
            got = [o.clone().view(torch.uint8) for o in outs]
            if ref is None:
                ref = got
            else:
                diff = [int((g != r).sum().item()) for g, r in zip(got, ref)]  # integer count (a float32 mean of 2e8 ones is not exact)
                print(f"check {tag}: cfg {c} vs cfg {cfgs[0]}: differing bytes per output {diff} of {[g.numel() for g in got]}", flush=True)
                assert all(d == 0 for d in diff), "outputs differ"
    if args.timeline:
        nwg = 256
        dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        _lib.call("fluxmi_gemm_debug_buffer", dbg.data_ptr())
        for _ in range(3):
            run(19)
        torch.cuda.synchronize()
        dbg.zero_()
        run(19)
        torch.cuda.synchronize()
        _lib.call("fluxmi_gemm_debug_buffer", None)
        d = dbg.cpu().view(nwg, 8, 8)
        used = d[:, :, 0] != 0
        kl = (d[:, :, 1] - d[:, :, 0])[used].float()
        ep = (d[:, :, 2] - d[:, :, 1])[used].float()
        gap = (d[:, 1:, 0] - d[:, :-1, 2])[used[:, 1:]].float()
        # the shader-clock counters of different XCDs are not aligned: spans and the clock are taken per workgroup (tile 0 start .. last
        # tile end against the 100 MHz real-time stamps of its first and last tile ends)
        ntile = used.sum(1)
        wg = torch.nonzero(ntile >= 2).flatten()
        last = (ntile[wg] - 1).clamp(max=7)
        cyc = d[wg, last, 2] - d[wg, 0, 2]
        rt = (d[wg, last, 3] - d[wg, 0, 3]).float() * 10e-9
        clk = (cyc.float() / rt)[rt > 0]
        span = (d[wg, last, 2] - d[wg, 0, 0]).float()
        print(f"timeline {tag}: {int(used.sum())} tiles on {int(used.any(1).sum())} workgroups; K loop {kl.mean():.0f} cycles (min {kl.min():.0f} max {kl.max():.0f}), "
              f"epilogue {ep.mean():.0f} (min {ep.min():.0f} max {ep.max():.0f}), tile-to-tile gap {gap.mean() if gap.numel() else 0:.0f}; "
              f"per-workgroup span {span.mean() if span.numel() else 0:.0f} cycles; shader clock {clk.median().item() / 1e9 if clk.numel() else float('nan'):.3f} GHz "
              f"(min {clk.min().item() / 1e9 if clk.numel() else float('nan'):.3f} max {clk.max().item() / 1e9 if clk.numel() else float('nan'):.3f})", flush=True)
        lutm = used & (d[:, :, 5] != 0)
        if lutm.any():
            ph = lambda a, b: (d[:, :, a] - d[:, :, b])[lutm].float().mean().item()
            print(f"   table tiles ({int(lutm.sum())}): K-loop end -> accumulators converted {ph(5, 1):.0f}, table landed + barrier {ph(6, 5):.0f}, "
                  f"gathers + transposition {ph(7, 6):.0f}, barrier + third K-step + stores {ph(2, 7):.0f} cycles", flush=True)
        for j in range(8):
            u = used[:, j]
            if u.any():
                print(f"   tile #{j}: {int(u.sum())} workgroups, K loop {(d[:, j, 1] - d[:, j, 0])[u].float().mean():.0f}, "
                      f"epilogue {(d[:, j, 2] - d[:, j, 1])[u].float().mean():.0f}", flush=True)
        return
    for c in cfgs:
        run(c)
    torch.cuda.synchronize()
    res = {c: [] for c in cfgs}
    for _ in range(args.rounds if args.ab else 1):
        for c in cfgs:
            res[c].append(timed(c, args.iters))
    Mtot = sum(Ms)
    for c in cfgs:
        tf = [2 * Mtot * N * K / t / 1e12 for t in res[c]]
        print(f"{tag} cfg={c}: median {statistics.median(tf):7.1f} TF/s  max {max(tf):7.1f}  ({statistics.median(res[c]) * 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
