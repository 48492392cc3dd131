#!/usr/bin/env python
"""Headline benchmark: denoise it/s of the Flux hot path on MI355X (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}]
      N > 1: one rank per GPU over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
      (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or plainly as `python bench.py --gpus N`: the script then re-launches
      itself under torch.distributed.run on 127.0.0.1 and passes rank 0's JSON line through.

One "step" = one pass of the hot path = one Flux.forward + Euler update of the denoise loop (reference flux_pipeline.py:641-651)
on one latent per GPU, inputs resident in HBM, replayed from the captured hipGraph.  Synthetic seeded request + random-init weights
of the Flux architecture (no checkpoint exists offline).  Setup (untimed, like the reference's compile() warm-up,
flux_pipeline.py:197-212): 13 calibrating steps that freeze the F8Linear input scales.

--config selects the BASELINE.json configuration (default 2 = the one the metric is quoted on):
  1  Flux-schnell 256x256, 1-step requests, bf16 flow (nn.Linear, no fp8): weight-stream bound -> roofline in GB/s of HBM
  2  Flux-dev 1024x1024, 28 steps, fp8 F8Linear (quantize_modulation) + bf16 flow                       [default]
  3  Flux-dev 768x768, quantize_modulation + quantize_flow_embedder_layers
  5  config 2 + a synthetic rank-16 LoRA on every attention / MLP linear, fused into the fp8 weights (scale 1.0) before timing
  4  Flux-dev 1024x1024, batch 8 (BASELINE.md section 2, 595 TFLOP per loop iteration): 8 / N images per GPU through ONE engine per GPU --
     `--config 4` alone = all eight on one GPU, `--config 4 --gpus 8` = one image per GPU (the same work as `--config 2 --gpus 8`)
N GPUs = N batch-sharded replicas (weak scaling); collectives: one RCCL broadcast of the T5/CLIP embeddings + noise before the
loop and, during calibration only, the per-layer amax MAX-reduction.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel of the configuration (the fp8 MX-MFMA GEMM; for config 1
the bf16 GEMM's weight stream) from HIP-event timings taken live in this process on the launch stream; `cpu_baseline` times the
reference's CPU flow path on the host cores for a bounded sample: the UNMODIFIED reference when /root/reference is present
(build container), otherwise the oracle port that is pinned bit-for-bit to it (oracle/gen_golden*.py).
  python bench.py --cpu-baseline-only [--config C]      runs only that leg (no GPU needed)
  python bench.py --no-pmc                               skips the live rocprofv3 PMC passes (HBM traffic, MFMA busy; default at N = 1 when
                                                         rocprofv3 is on PATH, ~2 min); `roofline.source` says where each number came from
  python bench.py --gpus 2 --backend gloo --dry-run      no GPU: the multi-rank control flow (rendezvous, broadcast, sharding, in-step amax
                                                         exchange, barriers, max-over-ranks timing, one JSON line) on a stub engine
  python bench.py --gpus N --preflight                   first-contact check of the N-rank plumbing, < 30 s, no model: process group, rank-id
                                                         all-reduce, ONE broadcast of a request-sized payload, the latent gather; one JSON line;
                                                         exit codes 10 rendezvous / 11 all-reduce / 12 broadcast / 13 gather (works with
                                                         --backend gloo --dry-run on CPU and with --single-rank-group on a one-GPU box)
`roofline.frac` is the IN-STEP figure: the same command runs one more graph-replayed request under `rocprofv3 --kernel-trace` (a child
process of this script, after the timed region) and prices the GEMM family by its kernel time inside the steady steps; `roofline.step`
splits the step into GEMM / attention / LayerNorm / other / gaps.  The isolated back-to-back probe of rounds 1 - 4 stays as
`roofline.probe_frac` (it flatters a kernel: warm Infinity Cache, lower clock).  --no-step-trace skips the child.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

H100_COMPILED = {2: 11.5, 5: 11.5, 3: 20.8}  # reference README.md:25,34 -- the only published numbers for this metric (other hardware)
FP8_PEAK_TFLOPS = 5000.0   # MI355X dense fp8 MFMA (MX-scaled K=128/64 opcodes), /opt/skills/guides/MI355X_MICROARCH.md
BF16_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    1: dict(name="Flux-schnell 256x256, 1-step requests, bf16 flow (no fp8)", schnell=True, height=256, width=256, txt_len=256, quant=None,
            steps_per_request=1, lora=False),
    2: dict(name="Flux-dev 1024x1024, fp8 F8Linear + bf16 flow", schnell=False, height=1024, width=1024, txt_len=512,
            quant=dict(modulation=True, embedders=False), steps_per_request=None, lora=False),
    3: dict(name="Flux-dev 768x768, quantize_modulation + quantize_flow_embedder_layers", schnell=False, height=768, width=768, txt_len=512,
            quant=dict(modulation=True, embedders=True), steps_per_request=None, lora=False),
    4: dict(name="Flux-dev 1024x1024, fp8 F8Linear + bf16 flow, batch 8", schnell=False, height=1024, width=1024, txt_len=512,
            quant=dict(modulation=True, embedders=False), steps_per_request=None, lora=False, batch_total=8),
    5: dict(name="Flux-dev 1024x1024 + rank-16 LoRA fused into the fp8 weights (scale 1.0)", schnell=False, height=1024, width=1024, txt_len=512,
            quant=dict(modulation=True, embedders=False), steps_per_request=None, lora=True),
}


def gemm_shapes(Li, Lt, H=3072, batch=1):
    """(name, Ms, N, K, launches/step, fused epilogue) of the six grouped Linear launches of one step: exactly what
    engine.hip issues in fused mode -- txt+img streams of a double block share one grouped launch, and so do the `batch` samples of a pass."""
    L, Hm = Li + Lt, 4 * H
    d, s1 = (Lt, Li) * batch, (L,) * batch
    return [
        ("double.qkv(txt+img)", d, 3 * H, H, 19, "bf16"), ("double.proj(txt+img)", d, H, H, 19, "gate_resid"),
        ("double.mlp0(txt+img)", d, Hm, H, 19, "gelu_quant"), ("double.mlp2(txt+img)", d, H, Hm, 19, "gate_resid"),
        ("single.linear1", s1, 3 * H + Hm, H, 38, "split"), ("single.linear2", s1, H, H + Hm, 38, "gate_resid"),
    ]


def linear_flops_per_step(Li, Lt, H=3072, batch=1):
    return sum(2.0 * sum(Ms) * N * K * cnt for _, Ms, N, K, cnt, _e in gemm_shapes(Li, Lt, H, batch))


def kernel_source_key():
    """content hash of the GEMM kernel sources + dispatcher: counter files collected for other sources are reported as stale (null)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "flux-fp8-api_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(("gemm", "common", "api")) and f.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measure_gemm_roofline(torch, ops, dev, Li, Lt, fp8=True, iters=10, batch=1):
    """Average duration of one Linear GEMM launch of the step: the six launch shapes WITH their fused epilogues (fp8 path: GELU+quantise,
    gate*y+x in place, qkv|mlp split, V^T and normalised + rotated K in the attention layout; bf16 path of config 1: plain bf16 outputs, what the unfused engine issues), weighted by their
    count per step; HIP events on the launch stream, random operands.  Returns (flops, algorithmic bytes, seconds) per launch + table."""
    from fluxmi import _lib

    one = torch.tensor(1.0, device=dev)
    H = 3072
    lut = ops.build_quant_lut(one, _lib.E5M2, act=1) if (fp8 and _lib.get_tuning().qlut) else None
    lut_ptr = lut.data_ptr() if lut is not None else None
    tot_t = tot_f = tot_b = 0.0
    n_launch, table = 0, []
    # the attention-layout outputs the engine fuses into the qkv / linear1 launches (fluxmi_tuning_t.fuse_kv: V^T, and K with QKNorm + RoPE)
    fuse_kv = _lib.get_tuning().fuse_kv if fp8 else 0
    L_all = Li + Lt
    Lp = (L_all + 63) // 64 * 64
    heads = H // 128
    for name, Ms, N, K, cnt, epi in gemm_shapes(Li, Lt, batch=batch):
        groups, keep = [], []
        nbytes = 0
        attn_out = fuse_kv >= 1 and epi in ("bf16", "split") and N >= 3 * H
        if attn_out:
            vt_t = torch.empty(H, Lp, dtype=torch.bfloat16, device=dev)
            k_t = torch.empty(heads, L_all, 128, dtype=torch.float16, device=dev)
            pe_t = torch.randn(L_all, 64, 2, device=dev).bfloat16()
            kn_t = (1 + 0.1 * torch.randn(128, device=dev)).bfloat16()
            keep += [vt_t, k_t, pe_t, kn_t]
        tok0 = 0
        for M in Ms:
            if fp8:
                a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2)
                w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
            else:
                a = torch.randn(M, K, device=dev).bfloat16()
                w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
                epi = "bf16"
            bias = torch.randn(N, device=dev).bfloat16()
            keep += [a, w, bias]
            kw = {}
            eb = 1 if fp8 else 2
            nbytes += (M * K + N * K) * eb  # operands in; outputs added below
            if epi == "bf16":
                o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                code = _lib.EPI_BF16
                nbytes += M * N * 2
            elif epi == "gelu_quant":
                o = torch.empty(M, N, dtype=torch.float8_e5m2, device=dev)
                code, kw = _lib.EPI_GELU_QUANT, dict(q_scale=one.data_ptr(), q_lut=lut_ptr)
                nbytes += M * N
            elif epi == "gate_resid":
                o = torch.randn(M, N, device=dev).bfloat16()  # residual stream, updated in place
                gate = torch.randn(N, device=dev).bfloat16()
                keep.append(gate)
                code, kw = _lib.EPI_GATE_RESID, dict(gate=gate.data_ptr(), resid=o.data_ptr(), ldr=N)
                nbytes += 2 * M * N * 2
            else:  # split: q|k|v bf16 to C, gelu(mlp) fp8 into the [attn | mlp] buffer at column H
                o = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev)
                o2 = torch.empty(M, 5 * H, dtype=torch.float8_e5m2, device=dev)
                keep.append(o2)
                code, kw = _lib.EPI_SPLIT, dict(C2=o2.data_ptr(), ldc2=5 * H, split_n=3 * H, c2_col0=H, q_scale=one.data_ptr(), q_lut=lut_ptr)
                nbytes += M * 3 * H * 2 + M * 4 * H
            if attn_out:
                last = tok0 + M == L_all
                kw.update(vt_out=vt_t.data_ptr(), vt_ld=Lp, tok0=tok0, vt_rows=(Lp - tok0) if last else M, kv_col0=H, heads=heads)
                if fuse_kv >= 2:
                    kw.update(k_out=k_t.data_ptr(), pe=pe_t.data_ptr(), k_norm=kn_t.data_ptr(), k_rows=L_all, k_f16=bool(_lib.get_tuning().attn_f16k))
                tok0 = (tok0 + M) % L_all  # the next sample of the pass starts over (the probe's samples share one set of K / V^T buffers)
            if fp8 and _lib.get_tuning().w_pairs and name.split("(")[0] in ("double.qkv", "double.mlp0", "double.mlp2", "single.linear1", "single.linear2"):
                wp = ops.pair_rows(w)  # the engine's row-pair copy of these weights (fluxmi_gemm_group_t.W_pairs)
                keep.append(wp)
                kw.update(W_pairs=wp.data_ptr())
            keep.append(o)
            groups.append(ops.make_group(a.data_ptr(), w.data_ptr(), bias.data_ptr(), one.data_ptr() if fp8 else None, one.data_ptr() if fp8 else None,
                                         o.data_ptr(), M, K, o.stride(0), **kw))
        fn = lambda: ops.gemm_grouped(groups, N, K, fp8, _lib.E5M2, code, -1)
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / iters
        f = 2.0 * sum(Ms) * N * K
        table.append({"launch": name, "epilogue": epi, "per_step": cnt, "us": round(t * 1e6, 1), "tflops": round(f / t / 1e12, 1),
                      "alg_gbs": round(nbytes / t / 1e9, 1)})
        tot_t += t * cnt
        tot_f += f * cnt
        tot_b += nbytes * cnt
        n_launch += cnt
        del groups, keep
    return tot_f / n_launch, tot_b / n_launch, tot_t / n_launch, table


def measure_attention(torch, ops, dev, L, iters=10, H=24, batch=1):
    """The second kernel of the step (57 launches): joint attention at the step's shape, bf16 MFMA, fp8 output, HIP events on the
    launch stream.  Reported beside the GEMM roofline; `peak` is the dense bf16 MFMA figure."""
    Lp = (L + 63) // 64 * 64
    q = torch.randn(batch, H, L, 128, device=dev).bfloat16()
    k = torch.randn(batch, H, L, 128, device=dev).bfloat16()
    from fluxmi import _lib

    f16k = bool(_lib.get_tuning().attn_f16k)  # what the engine launches
    if f16k:
        k = k.half()
    vt = torch.randn(batch, H, 128, Lp, device=dev).bfloat16()
    one = torch.tensor(1.0, device=dev)
    o8 = torch.empty(batch, L, H * 128, dtype=torch.float8_e5m2, device=dev)
    for _ in range(2):
        ops.attention(q, k, vt, q_scale0=one, out=o8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention(q, k, vt, q_scale0=one, out=o8)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    f = 4.0 * L * L * 128 * H * batch
    wgs = ((L + 255) // 256) * H * batch
    mode = _lib.get_tuning().attn_split if f16k else 0
    plan = ops.attention_plan(batch, L, H) if mode else None
    if plan and mode == 1 and not plan["thin"]:
        plan = None  # the default takes the balanced grid for thin last rounds only (include/fluxmi.h, attn_split)
    kern = "attention2_kernel (8 waves x 32 rows, skewed pipeline, deferred rescale" + (", folded, barrier between the MFMA groups)" if f16k else ")")
    note = f"{wgs} tasks of 256 query rows on 256 CUs = {wgs / 256:.2f} rounds"
    if plan:
        note += (f"; balanced grid: per XCD {plan['full_per_x']} whole tasks, the other {plan['n_per_x'] - plan['full_per_x']} as {len(plan['pieces'])} pieces of "
                 "their key range, fp32 log-sum-exp merge (fluxmi_attention_plan)")
    return {"kernel": kern + ", bf16/f16 MFMA 32x32x16, fp8 output", "per_step": 57, "probe_us": round(t * 1e6, 1),
            "probe_achieved": round(f / t / 1e12, 1), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "probe_frac": round(f / t / 1e12 / BF16_PEAK_TFLOPS, 4),
            "flops_per_launch": f, "note": note}


def pmc_file(kind, cfg_id):
    return os.path.join(ROOT, "profiles", f"r06_{kind}_config{cfg_id}.json")


def read_pmc(kind, cfg_id):
    """Counter values collected by `bench.py --pmc` (rocprofv3 PMC passes over tools/gemm_probe.py); null when the kernel sources
    changed since (the file is keyed by a content hash of csrc/gemm*, attention*, common.h, api.cpp)."""
    try:
        with open(pmc_file(kind, cfg_id)) as f:
            d = json.load(f)
        return d if d.get("source_key") == kernel_source_key() else None
    except Exception:
        return None


def pmc_dir(cfg_id):
    return os.path.join(ROOT, "gpurun_out", f"pmc_config{cfg_id}")


def collect_pmc(cfg_id, Li, Lt):
    """Live rocprofv3 PMC passes (separate --pmc runs, never combined with traces: MI355X_MICROARCH.md 'HBM' / 'rocprofv3 PMC slots') over
    tools/gemm_probe.py --cfg -1 (the production dispatch) for each of the step's GEMM shapes; raw CSVs under gpurun_out/pmc_config<id>/."""
    out_dir = pmc_dir(cfg_id)
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    t_start, budget_s = time.time(), float(os.environ.get("FLUXMI_BENCH_PMC_BUDGET_S", "480"))  # the counters must not eat the driver's time limit
    for name, Ms, N, K, cnt, epi in gemm_shapes(Li, Lt):
        tag = name.split("(")[0].replace(".", "_")
        for pi, counters in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"])):
            d = os.path.join(out_dir, f"{tag}_p{pi}")
            left = budget_s - (time.time() - t_start)
            if left < 20:
                return
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.join(ROOT, "tools", "gemm_probe.py"), "--shape", f"{sum(Ms)},{N},{K}",
                   *(["--groups", ",".join(str(m) for m in Ms)] if len(Ms) > 1 else []), "--cfg", "-1", "--iters", "4",
                   "--epi", {"bf16": "bf16", "gate_resid": "gate", "gelu_quant": "gelu", "split": "split"}[epi]] + (["--vt"] if epi in ("split", "bf16") else [])
            try:
                subprocess.run(cmd, env=env, cwd="/tmp", check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=min(150.0, left))
            except Exception:  # noqa
                pass


def summarize_pmc(cfg_id, Li, Lt):
    """raw CSVs -> profiles/r06_{traffic,mfma}_config<id>.json, keyed by the kernel-source hash.  HBM bytes per launch =
    (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 tallies a 128-B fabric read at 64 B: MI355X_MICROARCH.md 'HBM'); matrix-pipe busy fraction =
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  The first dispatch of every kernel (cold caches, lazy init) is
    dropped; the 256x256 kernels of a launch are summed with its 128x128 peel."""
    import csv
    import glob

    out_dir = pmc_dir(cfg_id)
    per_traffic, per_busy = {}, {}
    for name, Ms, N, K, cnt, epi in gemm_shapes(Li, Lt):
        tag = name.split("(")[0].replace(".", "_")
        vals = {}
        for pi in range(3):
            for path in glob.glob(os.path.join(out_dir, f"{tag}_p{pi}", "**", "*counter_collection.csv"), recursive=True):
                seen = {}
                with open(path) as f:
                    for row in csv.DictReader(f):
                        kn = row.get("Kernel_Name", "")
                        if "gemm" not in kn:
                            continue
                        key = (kn, row["Counter_Name"])
                        seen[key] = seen.get(key, 0) + 1
                        if seen[key] == 1:
                            continue  # first dispatch of this kernel
                        vals.setdefault(row["Counter_Name"], {}).setdefault(kn, []).append(float(row["Counter_Value"]))
        tot = lambda c: sum(sum(v) / len(v) for v in vals.get(c, {}).values()) if vals.get(c) else None
        fs, ws, ga, mb = tot("FETCH_SIZE"), tot("WRITE_SIZE"), tot("GRBM_GUI_ACTIVE"), tot("SQ_VALU_MFMA_BUSY_CYCLES")
        if fs is not None and ws is not None:
            per_traffic[name] = (2.0 * fs + ws) * 1024.0
        if ga and mb:
            per_busy[name] = mb / (ga / 8.0 * 1024.0)
    key = kernel_source_key()
    w = {name: cnt for name, _m, _n, _k, cnt, _e in gemm_shapes(Li, Lt)}
    how = "bench.py --pmc: rocprofv3 --kernel-trace --pmc <counters> -- python tools/gemm_probe.py --shape M,N,K --cfg -1 (one pass per counter set)"
    if per_traffic:
        tot_b = sum(per_traffic[n] * w[n] for n in per_traffic) / sum(w[n] for n in per_traffic)
        with open(pmc_file("traffic", cfg_id), "w") as f:
            json.dump({"source_key": key, "bytes_per_launch": tot_b, "per_launch": per_traffic, "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB", "how": how}, f, indent=1)
    if per_busy:
        with open(pmc_file("mfma", cfg_id), "w") as f:
            json.dump({"source_key": key, "per_launch": per_busy,
                       "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "how": how}, f, indent=1)
    return per_traffic, per_busy


# ---------------------------------------------------------------------------------------------------------------------------
# In-step kernel times: one more request of the same command under rocprofv3 --kernel-trace
# ---------------------------------------------------------------------------------------------------------------------------
def step_trace_dir(cfg_id):
    return os.path.join(ROOT, "gpurun_out", f"step_trace_config{cfg_id}")


def collect_step_trace(args, timeout_s=420.0):
    """Re-run THIS command's workload once more as a child under `rocprofv3 --kernel-trace` (kernel trace only: no counters, no API
    traces): same config, same steps / warmup, one timed request, no probes / PMC / CPU baseline.  Returns the directory or None."""
    d = step_trace_dir(args.config)
    import shutil

    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "-d", d, "-o", "step", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
           "--config", str(args.config), "--steps", str(args.steps), "--warmup", str(args.warmup), "--requests", "1", "--no-pmc", "--no-cpu-baseline",
           "--no-step-trace", "--no-probe"] + (["--no-graph"] if args.no_graph else []) + (["--depth", str(args.depth)] if args.depth is not None else [])
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "FLUXMI_BENCH_CHILD"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, env=env, cwd="/tmp", check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
    except Exception as ex:  # noqa: the trace must never take the measurement down
        print(f"bench: step trace failed: {ex}", file=sys.stderr)
        return None
    return d


FAMILIES = (("gemm", ("gemm", "splitk_reduce")), ("attention", ("attention",)), ("ln", ("ln_modulate",)))


def summarize_step_trace(path, header=""):
    """*kernel_trace.csv under `path` -> the steady steps of the LAST denoise request: the dispatches between the first and the last
    euler_kernel of that request (its first step also carries the request's set-up -- modulation table, txt_in, quantising tables -- and is
    dropped).  Returns {steps, wall_ms, kernel_ms, launches, gemm_ms / attention_ms / ln_ms / other_ms / gaps_ms (per step), per-kernel rows,
    text (the table committed under profiles/)} or None."""
    import csv
    import glob
    import re

    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(path) else [path]
    if not files:
        return None
    ks = []
    for f in files:
        with open(f) as fh:
            ks += [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(fh)]
    ks.sort(key=lambda k: k[1])
    eul = [i for i, k in enumerate(ks) if "euler_kernel" in k[0]]
    if len(eul) < 3:
        return None
    # the last request = the longest suffix of euler kernels with no calibration / table-building kernel in between
    setup = ("amax_kernel", "calib_update", "timestep_rows_kernel", "build_qlut")
    last = len(eul) - 1
    first = last
    while first > 0 and not any(any(t in ks[j][0] for t in setup) for j in range(eul[first - 1], eul[first])):
        first -= 1
    whole_requests = False
    if last - first < 1:
        # one-step requests (config 1): there is no step without the request's set-up -- it IS part of every step of such a workload.
        # Window = the last (up to 24) requests, everything between their euler kernels included
        first, whole_requests = max(1, last - 24), True
        while first < last and any(any(t in ks[j][0] for t in ("amax_kernel", "calib_update")) for j in range(eul[first - 1], eul[last])):
            first += 1
        if last - first < 1:
            return None
    t0, t1, steps = ks[eul[first]][2], ks[eul[last]][2], last - first

    def short(name):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        return name if len(name) <= 110 else name[:107] + "..."

    rows, fam = {}, {f: 0.0 for f, _ in FAMILIES}
    fam["other"] = 0.0
    for n, s_, e in ks:
        if s_ >= t0 and e <= t1:
            k = short(n)
            c, t = rows.get(k, (0, 0.0))
            rows[k] = (c + 1, t + (e - s_))
            for f, keys in FAMILIES:
                if any(x in n for x in keys):
                    fam[f] += e - s_
                    break
            else:
                fam["other"] += e - s_
    wall = (t1 - t0) / steps / 1e6
    tot = sum(t for _, t in rows.values())
    out = {"steps": steps, "wall_ms": round(wall, 3), "kernel_ms": round(tot / steps / 1e6, 3), "launches": round(sum(c for c, _ in rows.values()) / steps, 1)}
    for f in fam:
        out[f + "_ms"] = round(fam[f] / steps / 1e6, 3)
    out["gaps_ms"] = round(wall - tot / steps / 1e6, 3)
    out["gemm_launches"] = round(sum(c for k, (c, _) in rows.items() if any(x in k for x in FAMILIES[0][1])) / steps, 1)
    out["attention_us"] = round(fam["attention"] / max(1, sum(c for k, (c, _) in rows.items() if "attention" in k)) / 1e3, 2)
    lines = ([header] if header else []) + [
        f"# steady state: {steps} graph-replayed denoise steps " + ("= one-step requests, each with its set-up (modulation table, txt_in)" if whole_requests else "of the last request") + f", wall {wall:.3f} ms/step, kernel time {tot / steps / 1e6:.3f} ms/step, "
        f"{out['launches']:.0f} launches/step; per step: GEMM family {out['gemm_ms']:.3f} ms, attention {out['attention_ms']:.3f}, LayerNorm+modulate "
        f"{out['ln_ms']:.3f}, other {out['other_ms']:.3f}, gaps {out['gaps_ms']:.3f}; columns are totals over those steps (ns resolution -> us)",
        f"{'kernel':112s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}"]
    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:112s} {c:7d} {t / 1e3:12.1f} {t / c / 1e3:10.2f} {100 * t / tot:6.2f}")
    out["text"] = "\n".join(lines) + "\n"
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# --pmc-step: hardware counters of the kernels INSIDE the step (opt-in: three more child runs, ~1 min each)
# ---------------------------------------------------------------------------------------------------------------------------
STEP_PMC_PASSES = (["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"])  # FETCH_SIZE and WRITE_SIZE do not fit one pass


def step_source_key():
    """content hash of every kernel source (the in-step counters cover all of them)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "flux-fp8-api_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def step_pmc_file(cfg_id):
    return os.path.join(ROOT, "profiles", f"r06_step_pmc_config{cfg_id}.json")


def collect_step_pmc(args, timeout_s=300.0):
    """The workload once more per counter set, EAGER launches (one AQL dispatch per kernel, what the counter collection serialises anyway), a
    short request: `rocprofv3 --kernel-trace --pmc <set>` -- counters in their own runs, never combined with API traces."""
    import shutil

    root = os.path.join(ROOT, "gpurun_out", f"step_pmc_config{args.config}")
    shutil.rmtree(root, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "FLUXMI_BENCH_CHILD"):
        env.pop(k, None)
    for pi, counters in enumerate(STEP_PMC_PASSES):
        d = os.path.join(root, f"p{pi}")
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
               "--config", str(args.config), "--steps", "6" if CONFIGS[args.config]["steps_per_request"] is None else "8", "--warmup", "2", "--requests", "1",
               "--no-graph", "--no-pmc", "--no-cpu-baseline", "--no-step-trace", "--no-probe"] + (["--depth", str(args.depth)] if args.depth is not None else [])
        try:
            subprocess.run(cmd, env=env, cwd="/tmp", check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
        except Exception as ex:  # noqa
            print(f"bench: step PMC pass {counters} failed: {ex}", file=sys.stderr)
    return root


def summarize_step_pmc(root, cfg_id, ms_per_step=None):
    """counter_collection.csv of the three passes -> per kernel family, per steady step of the last request: HBM-side bytes
    ((2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 tallies a 128-byte fabric read at 64 B, MI355X_MICROARCH.md 'HBM'; Infinity-Cache hits count)
    and the matrix-pipe busy fraction SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  Written to profiles/ with the
    kernel-source hash."""
    import csv
    import glob

    tot = {}
    steps_seen = None
    for pi, counters in enumerate(STEP_PMC_PASSES):
        rows = {}
        for path in glob.glob(os.path.join(root, f"p{pi}", "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], {}])[1][r["Counter_Name"]] = float(r["Counter_Value"])
        seq = [rows[k] for k in sorted(rows)]
        eul = [i for i, (n, _c) in enumerate(seq) if "euler_kernel" in n]
        if len(eul) < 2:
            continue
        setup = ("amax_kernel", "calib_update", "timestep_rows_kernel", "build_qlut")
        last = len(eul) - 1
        first = last
        while first > 0 and not any(any(t in seq[j][0] for t in setup) for j in range(eul[first - 1], eul[first])):
            first -= 1
        if last - first < 1:
            first = max(1, last - 6)  # one-step requests: whole requests, set-up included
        steps = last - first
        steps_seen = steps if steps_seen is None else min(steps_seen, steps)
        for n, c in seq[eul[first] + 1: eul[last] + 1]:
            fam = next((f for f, keys in FAMILIES if any(x in n for x in keys)), "other")
            for cn, v in c.items():
                tot.setdefault(fam, {}).setdefault(cn, 0.0)
                tot[fam][cn] += v / steps
    if not tot or steps_seen is None:
        return None
    out = {"source_key": step_source_key(), "steps": steps_seen,
           "how": "bench.py --pmc-step: rocprofv3 --kernel-trace --pmc <set> -- python bench.py --no-graph --steps 6 ... (one child run per counter set), steady steps of the last request",
           "formula": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "per_step": {}}
    all_b = 0.0
    all_busy = [0.0, 0.0]
    for fam, c in tot.items():
        e = {}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            e["fetch_bytes"], e["write_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
            all_b += e["hbm_bytes"]
        if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
            all_busy[0] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
            all_busy[1] += c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        out["per_step"][fam] = e
    out["hbm_bytes_per_step"] = all_b
    out["mfma_busy_frac_step"] = round(all_busy[0] / all_busy[1], 4) if all_busy[1] else None
    if ms_per_step:
        out["hbm_gbs_at_timed_ms_per_step"] = round(all_b / (ms_per_step * 1e-3) / 1e9, 1)
        out["ms_per_step_timed"] = ms_per_step
    with open(step_pmc_file(cfg_id), "w") as f:
        json.dump(out, f, indent=1)
    return out


def read_step_pmc(cfg_id):
    try:
        with open(step_pmc_file(cfg_id)) as f:
            d = json.load(f)
        return d if d.get("source_key") == step_source_key() else None
    except Exception:  # noqa
        return None


# ---------------------------------------------------------------------------------------------------------------------------
# --preflight: first contact of the N-rank plumbing (no model, < 30 s)
# ---------------------------------------------------------------------------------------------------------------------------
PREFLIGHT_CODES = {"rendezvous": 10, "allreduce": 11, "broadcast": 12, "gather": 13}


def preflight(args):
    """process group -> rank-id all-reduce -> ONE broadcast_request of the real payload size (T5 states, CLIP vector, packed noise of
    one image per rank at the config's resolution) -> gather_latents -> one JSON line from rank 0.  Every stage has its own exit code."""
    t_all = time.time()
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # RCCL prints banners on fd 1
    import torch
    import torch.distributed as td

    from fluxmi import dist as fdist

    C = CONFIGS[args.config]

    def fail(stage, msg):
        print(json.dumps({"preflight": "failed", "stage": stage, "error": msg, "n_gpus": args.gpus}), file=sys.stderr, flush=True)
        os._exit(PREFLIGHT_CODES[stage])

    def inject(stage):  # test hook: FLUXMI_PREFLIGHT_FAIL=<stage>[:<rank>] breaks that stage (on that rank)
        want = os.environ.get("FLUXMI_PREFLIGHT_FAIL", "").split(":")
        if want[0] == stage and (len(want) < 2 or want[1] == os.environ.get("RANK", "0")):
            raise RuntimeError("injected failure")

    times = {}
    try:
        t0 = time.time()
        inject("rendezvous")
        rank, world, local = fdist.init_from_env("gloo" if args.dry_run and args.backend != "nccl" else args.backend)
        if args.single_rank_group and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29519")
            if args.backend == "nccl" and not args.dry_run:
                torch.cuda.set_device(0)
            td.init_process_group(backend=args.backend, rank=0, world_size=1)
        if world != args.gpus:
            raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        if not td.is_initialized():
            raise RuntimeError("no process group (N = 1 needs --single-rank-group)")
        if not args.dry_run:
            torch.cuda.set_device(local)
        dev = torch.device("cpu") if args.dry_run else torch.device("cuda", local)
        times["init_s"] = round(time.time() - t0, 2)
    except Exception as ex:  # noqa
        fail("rendezvous", f"{type(ex).__name__}: {ex}")
    sync = (lambda: None) if args.dry_run else torch.cuda.synchronize
    try:
        t0 = time.time()
        inject("allreduce")
        rid = torch.tensor([float(rank), 1.0], device=dev, dtype=torch.float64)
        td.all_reduce(rid, op=td.ReduceOp.SUM)
        sync()
        if int(round(rid[1].item())) != world or int(round(rid[0].item())) != world * (world - 1) // 2:
            raise RuntimeError(f"{int(round(rid[1].item()))} ranks answered (id sum {int(round(rid[0].item()))}), expected {world}")
        times["allreduce_s"] = round(time.time() - t0, 3)
    except Exception as ex:  # noqa
        fail("allreduce", f"rank {rank}: {type(ex).__name__}: {ex}")
    B = world * (C.get("batch_total", world) // world)
    Li, Lt = (C["height"] // 16) * (C["width"] // 16), C["txt_len"]
    try:
        t0 = time.time()
        g = torch.Generator().manual_seed(7)
        want = [torch.randn(B, Lt, 4096, generator=g).bfloat16(), torch.randn(B, 768, generator=g).bfloat16(), torch.randn(B, Li, 64, generator=g).bfloat16()]
        parts = [w.to(dev) if rank == 0 else torch.zeros_like(w, device=dev) for w in want]
        got = fdist.broadcast_request(*parts, src=0)
        sync()
        inject("broadcast")
        for w, t in zip(want, got):
            if not torch.equal(t.cpu().view(torch.int16), w.view(torch.int16)):
                raise RuntimeError("payload differs from the source rank's")
        times["broadcast_s"] = round(time.time() - t0, 3)
        times["broadcast_bytes"] = sum(w.numel() * 2 for w in want)
    except Exception as ex:  # noqa
        fail("broadcast", f"rank {rank}: {type(ex).__name__}: {ex}")
    try:
        t0 = time.time()
        lo, hi = fdist.shard_bounds(B, rank, world)
        lat = torch.stack([torch.full((Li, 64), float(i + 1)) for i in range(lo, hi)]).bfloat16().to(dev) if hi > lo else torch.zeros(0, Li, 64, dtype=torch.bfloat16, device=dev)
        allv = fdist.gather_latents(lat, B, dst=0)
        sync()
        inject("gather")
        if rank == 0 and not (allv.shape[0] == B and all(float(allv[i, 0, 0]) == float(i + 1) and float(allv[i, -1, -1]) == float(i + 1) for i in range(B))):
            raise RuntimeError("gathered latents are not the ranks' shards in order")
        times["gather_s"] = round(time.time() - t0, 3)
    except Exception as ex:  # noqa
        fail("gather", f"rank {rank}: {type(ex).__name__}: {ex}")
    td.barrier()
    backend, nranks = td.get_backend(), td.get_world_size()
    td.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps({"preflight": "ok", "n_gpus": world, "nranks": nranks, "backend": backend, "batch": B, "config": args.config,
                                           "total_s": round(time.time() - t_all, 2), **times}) + "\n").encode())


# ---------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg_id):
    """The reference's CPU flow path on the host cores, bounded sample: runs oracle/cpu_baseline.py in its own interpreter (the
    unmodified reference's `modules` / `util` packages shadow this repo's same-named host modules, so the two cannot share a process)."""
    C = CONFIGS[cfg_id]
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--height", str(C["height"]), "--width", str(C["width"]),
           "--txt-len", str(C["txt_len"]), "--full-step"] + (["--schnell"] if C["schnell"] else [])
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=900).stdout.strip().splitlines()[-1]
    return json.loads(out)


def synthetic_lora(p, rank=16, seed=5):
    """rank-16 LoRA on every attention / MLP linear of the model (BFL-dotted keys, what Flux.load_lora takes as a dict;
    lora_loading.py:608-612): fused qkv layers get the 'uneven rank' form A [3r, K], B [3N', r]."""
    import torch

    g = torch.Generator().manual_seed(seed)
    H, Hm = p.hidden_size, int(p.hidden_size * p.mlp_ratio)
    lora = {}

    def add(name, N, K, uneven=False):
        lora[name + ".lora_A.weight"] = torch.randn((3 if uneven else 1) * rank, K, generator=g) * 0.02
        lora[name + ".lora_B.weight"] = torch.randn(N, rank, generator=g) * 0.02

    for i in range(p.depth):
        for s in ("img", "txt"):
            add(f"double_blocks.{i}.{s}_attn.qkv", 3 * H, H, True)
            add(f"double_blocks.{i}.{s}_attn.proj", H, H)
            add(f"double_blocks.{i}.{s}_mlp.0", Hm, H)
            add(f"double_blocks.{i}.{s}_mlp.2", H, Hm)
    for i in range(p.depth_single_blocks):
        add(f"single_blocks.{i}.linear1", 3 * H + Hm, H)
        add(f"single_blocks.{i}.linear2", H, H + Hm)
    return lora


def _clock_from_samples(c):
    """[R + 1][8 blocks][xcc id, s_memtime, s_memrealtime] -> per repeat the average shader clock in GHz (None if the counters do not
    behave like a shader-clock / 100 MHz pair).  Block b of a sample runs on some XCD; consecutive samples are paired by XCC id because
    the shader-clock counters of different XCDs are not aligned."""
    out = []
    for r in range(c.shape[0] - 1):
        per_x = []
        a = {int(c[r, b, 0]): (int(c[r, b, 1]), int(c[r, b, 2])) for b in range(8)}
        for b in range(8):
            x, t1, rt1 = int(c[r + 1, b, 0]), int(c[r + 1, b, 1]), int(c[r + 1, b, 2])
            if x in a and rt1 > a[x][1] and t1 > a[x][0]:
                per_x.append((t1 - a[x][0]) / (rt1 - a[x][1]) * 0.1)  # cycles per 10 ns tick -> GHz
        per_x.sort()
        out.append(round(per_x[len(per_x) // 2], 3) if per_x else None)
    return out


def _pci_bdf(torch, dev):
    """PCI address of a torch device as sysfs spells it, or None"""
    try:
        p = torch.cuda.get_device_properties(dev)
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa
        return None


class _PowerSampler:
    """Board power (and sclk, when hwmon exposes it) sampled from sysfs every 20 ms while a timed repeat runs.  Best effort: absent
    files -> summary() is None.  Reading sysfs from a host thread does not touch the GPU queue.
    `pci_bdf` ("0000:c1:00.0") = the device the timed work runs on: its hwmon directory is the only trustworthy one -- on a multi-GPU node
    whose other GPUs are hidden from the process, /sys/class/drm/card0 is usually ANOTHER GPU (round 4 read 245 / 320 / 635 W from such
    neighbours next to 1258 - 1277 W from the right device).  Without a match the first hwmon found is used and the summary says so."""

    def __init__(self, pci_bdf=None, sysfs_root="/sys"):
        import glob
        import threading

        self._threading = threading
        self.matched = False
        cands = []
        if pci_bdf:
            base = os.path.join(sysfs_root, "bus", "pci", "devices", pci_bdf.lower(), "hwmon", "hwmon*")
            cands = sorted(glob.glob(os.path.join(base, "power1_average")) + glob.glob(os.path.join(base, "power1_input")))
            self.matched = bool(cands)
        if not cands:
            base = os.path.join(sysfs_root, "class", "drm", "card*", "device", "hwmon", "hwmon*")
            cands = sorted(glob.glob(os.path.join(base, "power1_average")) + glob.glob(os.path.join(base, "power1_input")))
            if pci_bdf:  # a card whose device link resolves to the wanted PCI address
                for c in cands:
                    dev = os.path.realpath(os.path.join(os.path.dirname(c), "..", ".."))
                    if os.path.basename(dev).lower() == pci_bdf.lower():
                        cands, self.matched = [c], True
                        break
        self.pfile = cands[0] if cands else None
        f = sorted(glob.glob(os.path.dirname(self.pfile) + "/freq1_input")) if self.pfile else []
        self.ffile = f[0] if f else None
        self.watts, self.mhz, self._run, self._th = [], [], False, None

    def _loop(self):
        while self._run:
            try:
                with open(self.pfile) as fh:
                    self.watts.append(int(fh.read().strip()) / 1e6)
                if self.ffile:
                    with open(self.ffile) as fh:
                        self.mhz.append(int(fh.read().strip()) / 1e6)
            except (OSError, ValueError):
                pass
            time.sleep(0.02)

    def start(self):
        if not self.pfile:
            return
        self._run = True
        self._th = self._threading.Thread(target=self._loop, daemon=True)
        self._th.start()

    def stop(self):
        self._run = False
        if self._th:
            self._th.join(timeout=1.0)
            self._th = None

    def summary(self):
        if not self.watts:
            return None
        d = {"source": self.pfile, "device_matched": self.matched, "samples": len(self.watts), "watts_mean": round(sum(self.watts) / len(self.watts), 1),
             "watts_max": round(max(self.watts), 1)}
        if not self.matched:
            d["note"] = "hwmon directory not matched to the PCI address of the timed device: may belong to another GPU of the node"

        if self.mhz:
            d["hwmon_sclk_mhz_mean"] = round(sum(self.mhz) / len(self.mhz), 1)
        return d


class _StubFlux:
    """--dry-run: CPU stand-in with the surface of modules.flux_model.Flux that this script drives (denoise, calibration state, amax
    exchange).  Per-sample arithmetic only, like the real model, so sharded == whole-batch; every calibrating step issues one
    reduction through the installed exchange, like the engine's per-layer hook does."""

    def __init__(self, has_f8=True):
        self._xchg, self._frozen, self._trial, self._has_f8 = None, not has_f8, 0, has_f8
        self._engine, self.exchanges = None, 0

    def enable_amax_exchange(self, reduce_fn=None):
        import torch.distributed as td

        self._xchg = None if reduce_fn is False else (reduce_fn or (lambda t: td.all_reduce(t, op=td.ReduceOp.MAX)))

    def calibration_state(self):
        return (self._frozen, self._trial) if self._has_f8 else (None, 0)

    def f8_modules(self):
        return []

    def load_lora(self, *a, **k):
        pass

    def denoise(self, img, img_ids, txt, txt_ids, vec, ts, guidance=3.5, use_graph=True):
        import torch

        x = img.float().clone()
        ctx = txt.float().mean(dim=(1, 2)).reshape(-1, 1, 1) + vec.float().mean(dim=1).reshape(-1, 1, 1)
        for t0, t1 in zip(ts[:-1], ts[1:]):
            if not self._frozen:
                if self._xchg is not None:
                    a = x.abs().amax().reshape(1)
                    self._xchg(a)
                    self.exchanges += 1
                self._trial += 1
                self._frozen = self._trial > 12
            x = x + (t1 - t0) * (0.1 * x + ctx)
        return x.to(img.dtype)


def _self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: re-launch under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1) and pass rank 0's JSON line through."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, FLUXMI_BENCH_CHILD="1")
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=28)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--pmc", action="store_true", help="force the live rocprofv3 PMC passes (default: on at N = 1 when rocprofv3 is on PATH)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC passes (~2 min)")
    ap.add_argument("--pmc-summarize", action="store_true", help="no GPU: rebuild profiles/r06_*_config<id>.json from gpurun_out/pmc_config<id>/")
    ap.add_argument("--requests", type=int, default=3, help="timed repeats of the K steps (each bracketed and timed on its own; the median is reported)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: run the multi-rank control flow on a stub engine (CPU tensors)")
    ap.add_argument("--depth", type=int, default=None, help="debug only: fewer blocks (the result is then flagged invalid)")
    ap.add_argument("--preflight", action="store_true", help="check the N-rank plumbing (group, all-reduce, request broadcast, latent gather) and exit; no model")
    ap.add_argument("--no-step-trace", action="store_true", help="skip the extra request under rocprofv3 --kernel-trace (roofline.frac is then the isolated probe's)")
    ap.add_argument("--pmc-step", action="store_true", help="also collect hardware counters of the kernels INSIDE the step (HBM-side bytes, matrix-pipe busy): three more "
                    "child runs under rocprofv3 --pmc, ~1 min each; the result is kept in profiles/r06_step_pmc_config<id>.json and reported while the kernel sources match")
    ap.add_argument("--keep-trace", action="store_true", help="keep the raw rocprofv3 CSVs of the step trace under gpurun_out/step_trace_config<id>/")
    ap.add_argument("--no-probe", action="store_true", help="skip the isolated GEMM / attention probe loops (used by the step-trace child)")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="N = 1 only: create a one-rank process group anyway and run every collective of the N > 1 path through it (broadcast, "
                         "in-step amax all-reduce from the engine's hook, barriers): exercises the RCCL plumbing on a one-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="N > 1 on a ONE-GPU box (with --backend gloo): every rank builds its real engine on cuda:0 and the whole N-rank timed path runs -- "
                         "request broadcast, in-step amax exchange from the engine's hook, barriers, max-over-ranks timing, one JSON line.  The ranks share "
                         "the chip, so the line is flagged `share_device` and is NOT a scaling measurement")
    args = ap.parse_args()
    C = CONFIGS[args.config]
    if args.share_device and (args.backend != "gloo" or args.gpus < 2):
        raise SystemExit("--share-device needs --gpus N >= 2 and --backend gloo (RCCL refuses two ranks on one device)")

    if args.pmc_summarize:
        Li, Lt = (C["height"] // 16) * (C["width"] // 16), C["txt_len"]
        print(json.dumps(summarize_pmc(args.config, Li, Lt), indent=1))
        return
    if args.cpu_baseline_only:
        print(json.dumps({"config": args.config, "workload": C["name"], "cpu_baseline": cpu_baseline(args.config)}), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        _self_launch(args)
    if args.preflight:
        preflight(args)
        return
    # stdout carries exactly ONE line, the JSON: everything else any library writes to fd 1 (RCCL prints a version banner there when a
    # communicator is created or destroyed) goes to stderr; rank 0 writes the line to the saved descriptor at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    emit = lambda line: os.write(real_stdout, (line + "\n").encode())

    import torch
    import torch.distributed as td

    from fluxmi import dist as fdist

    rank, world, local = fdist.init_from_env("gloo" if args.dry_run and args.backend != "nccl" else args.backend)
    group1 = args.single_rank_group and world == 1
    if group1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl" and not args.dry_run:
            torch.cuda.set_device(0)
        td.init_process_group(backend=args.backend, rank=0, world_size=1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or without a torchrun environment: the script then launches its own ranks)")
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        sync = lambda: None
    else:
        if args.share_device:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize

    from fluxmi import synth

    if dry:
        from types import SimpleNamespace

        p = SimpleNamespace(in_channels=64, vec_in_dim=64, context_in_dim=128, hidden_size=256, mlp_ratio=4.0, num_heads=2, depth=2,
                            depth_single_blocks=2, guidance_embed=not C["schnell"], qkv_bias=True)
        _lib = ops = None
    else:
        import util
        from float8_quantize import quantize_flow_transformer_and_dispatch_float8
        from fluxmi import _lib, ops

        cfg = util.load_config(util.ModelVersion.flux_schnell if C["schnell"] else util.ModelVersion.flux_dev, flow_dtype="bfloat16",
                               quantize_modulation=bool(C["quant"] and C["quant"]["modulation"]),
                               quantize_flow_embedder_layers=bool(C["quant"] and C["quant"]["embedders"]))
        if args.depth is not None:
            cfg.params.depth, cfg.params.depth_single_blocks = args.depth, 2 * args.depth
        p = cfg.params
    if C.get("batch_total", world) % world:
        raise SystemExit(f"--config {args.config} shards {C['batch_total']} images: --gpus must divide it")
    ipg = C.get("batch_total", world) // world           # images per GPU (config 4: 8 / N; every other config: one)
    spr = C["steps_per_request"] or args.steps          # steps per denoise request (config 1: 1-step requests)
    n_req = max(1, args.steps // spr)
    t_setup = time.time()
    with torch.inference_mode():
        if dry:
            model = _StubFlux(has_f8=C["quant"] is not None)
        else:
            sd = synth.make_state_dict(p, seed=0, device=dev)
            model = util.load_flow_model(cfg, sd)
            del sd
            if C["quant"] is not None:
                quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                              quantize_modulation=C["quant"]["modulation"],
                                                              quantize_flow_embedder_layers=C["quant"]["embedders"])
            else:
                model.to(dev)
            torch.cuda.empty_cache()
        # request: rank 0 plays the text-encoder rank; ONE RCCL broadcast of embeddings + noise
        hw = (64, 64, 32) if dry else (C["height"], C["width"], C["txt_len"])
        inp = synth.make_inputs(p, hw[0], hw[1], hw[2], batch=world * ipg, seed=0)
        txt, vec, img = (inp[k].to(dev) for k in ("txt", "y", "img"))
        if world > 1 or group1:
            if rank != 0:
                txt, vec, img = torch.zeros_like(txt), torch.zeros_like(vec), torch.zeros_like(img)
            txt, vec, img = fdist.broadcast_request(txt, vec, img, src=0)
        lo, hi = fdist.shard_bounds(world * ipg, rank, world)
        txt, vec, img = txt[lo:hi].contiguous(), vec[lo:hi].contiguous(), img[lo:hi].contiguous()
        img_ids, txt_ids = inp["img_ids"][lo:hi].to(dev), inp["txt_ids"][lo:hi].to(dev)
        Li, Lt = img.shape[1], txt.shape[1]
        sched = lambda n: util_schedule(n, Li, shift=not C["schnell"])
        nranks = td.get_world_size() if (world > 1 or group1) else 1
        backend = td.get_backend() if (world > 1 or group1) else None
        calibration = None
        if C["quant"] is not None:
            # calibration (untimed): 13 unfused steps freeze every F8Linear input scale.  Batch-sharded replicas MAX-reduce each layer's
            # amax inside every calibrating step (float8_quantize.py:227 takes it over the whole batch).  A rank whose exchange fails
            # cannot rejoin the others' collectives: it reports and exits non-zero (torch.distributed.run then tears the job down) --
            # there is no silent per-rank fallback.  FLUXMI_BENCH_AMAX_XCHG=0 selects the coarser scheme on purpose: per-rank
            # calibration, then ONE all-reduce of the running-amax trials (fluxmi.dist.sync_calibration).
            xchg = (world > 1 or group1) and os.environ.get("FLUXMI_BENCH_AMAX_XCHG", "1") != "0"
            calibration = "in-step per-layer amax all-reduce(MAX)" if xchg else ("per-rank + one all-reduce of the trials" if world > 1 else "single rank")
            try:
                if xchg:
                    model.enable_amax_exchange()
                if dry and os.environ.get("FLUXMI_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank whose exchange breaks
                    raise RuntimeError("injected exchange failure")
                model.denoise(img, img_ids, txt, txt_ids, vec, sched(13), guidance=3.5, use_graph=False)
                sync()
            except Exception as e:
                print(json.dumps({"error": f"rank {rank}: calibration failed ({type(e).__name__}: {e})", "calibration": calibration,
                                  "n_gpus": world, "backend": backend}), file=sys.stderr, flush=True)
                os._exit(3)
            if xchg:
                model.enable_amax_exchange(False)
            elif world > 1 and not dry:
                fdist.sync_calibration(model.f8_modules(), model)
            assert model.calibration_state()[0]
        lora_s = rebuild_ms = None
        if C["lora"] and not dry:
            t0 = time.time()
            model.load_lora(synthetic_lora(p), 1.0, name="bench-rank16")
            sync()
            lora_s = time.time() - t0
            # what the engine redoes on the first launch after the rebind that follows the fuse: the row-pair copies of the fused weights (8 GB at
            # Flux-dev) and the 77 quantising-epilogue tables -- first eager step minus second eager step
            ev = []
            for _ in range(2):
                t0 = time.perf_counter()
                model.denoise(img, img_ids, txt, txt_ids, vec, sched(1), guidance=3.5, use_graph=False)
                sync()
                ev.append(time.perf_counter() - t0)
            rebuild_ms = round(max(0.0, ev[0] - ev[1]) * 1e3, 1)
        if args.warmup > 0:
            model.denoise(img, img_ids, txt, txt_ids, vec, sched(max(min(args.warmup, spr), 2) if spr > 1 else 1), guidance=3.5,
                          use_graph=not args.no_graph)
            if spr == 1:
                for _ in range(max(args.warmup - 1, 1)):
                    model.denoise(img, img_ids, txt, txt_ids, vec, sched(1), guidance=3.5, use_graph=not args.no_graph)
        sync()
        setup_s = time.time() - t_setup

        ts = sched(spr)
        # ---- timed region: R back-to-back repeats of the SAME K steps, each bracketed by barrier + synchronize on both sides and timed
        # on its own; the line reports the MEDIAN repeat (">= 3 full graph replays, median") and every repeat beside it.
        # Around each repeat one tiny kernel samples the shader-clock counter and the 100 MHz real-time counter on every XCD
        # (fluxmi_clock_sample): their ratio is the clock the chip SUSTAINED over that repeat -- this workload is power-limited, and a
        # slow box or a throttling one shows up here instead of as unexplained spread.  Board power / sclk from hwmon, when readable.
        R = max(1, args.requests)
        clk = None if dry else torch.zeros((R + 1) * 8 * 3, dtype=torch.int64, device=dev)
        power = _PowerSampler(_pci_bdf(torch, dev)) if (not dry and rank == 0) else None
        elapsed_each, finite = [], True
        out = None
        for rep in range(R):
            if world > 1 or group1:
                td.barrier()
            sync()
            if clk is not None and rep == 0:
                _lib.call("fluxmi_clock_sample", clk.data_ptr(), ops._stream())
                sync()
            if power:
                power.start()
            t0 = time.perf_counter()
            for _ in range(n_req):
                out = model.denoise(img, img_ids, txt, txt_ids, vec, ts, guidance=3.5, use_graph=not args.no_graph)
            sync()
            if world > 1 or group1:
                td.barrier()
            el = time.perf_counter() - t0
            if power:
                power.stop()
            if clk is not None:
                _lib.call("fluxmi_clock_sample", clk.data_ptr() + (rep + 1) * 8 * 3 * 8, ops._stream())
                sync()
            finite = finite and bool(torch.isfinite(out.float()).all())
            if world > 1 or group1:
                tt = torch.tensor([el, 0.0 if finite else 1.0], device=dev, dtype=torch.float64)
                td.all_reduce(tt, op=td.ReduceOp.MAX)
                el, finite = float(tt[0].item()), float(tt[1].item()) == 0.0
            elapsed_each.append(el)
        elapsed = sorted(elapsed_each)[len(elapsed_each) // 2] if len(elapsed_each) % 2 else sorted(elapsed_each)[len(elapsed_each) // 2 - 1]
        ms_ev_v, n_ev_v = 0.0, 0
        if not dry:
            # engine-side meter: hipEvents recorded on the launch stream around the graph replays of the LAST request
            ms_ev, n_ev = _lib.C.c_float(0), _lib.C.c_int(0)
            _lib.call("fluxmi_engine_last_timing", model._engine, _lib.C.byref(ms_ev), _lib.C.byref(n_ev))
            ms_ev_v, n_ev_v = ms_ev.value, n_ev.value
        clock_each = _clock_from_samples(clk.cpu().view(R + 1, 8, 3)) if clk is not None else None
        if world > 1 or group1:
            # nranks is what the process group REPORTS after a collective over every rank (sum of rank ids): a silently degraded group
            # (a rank that fell back to a private communicator) shows up here, not as a wrong throughput
            rid = torch.tensor([float(rank), 1.0], device=dev, dtype=torch.float64)
            td.all_reduce(rid, op=td.ReduceOp.SUM)
            nranks_seen, id_sum = int(round(rid[1].item())), int(round(rid[0].item()))
            if nranks_seen != world or id_sum != world * (world - 1) // 2:
                print(json.dumps({"error": f"rank {rank}: process group degraded: {nranks_seen} ranks answered (id sum {id_sum}), expected {world}"}),
                      file=sys.stderr, flush=True)
                os._exit(4)
            nranks = nranks_seen

        result = None
        if rank == 0:
            steps_done = n_req * spr
            ms_per_step = elapsed / steps_done * 1e3
            loop_its = steps_done / elapsed                 # loop iterations per second on one GPU (what the reference's tqdm prints)
            its = world * ipg * loop_its                    # whole job: every image of every GPU advances one step per loop iteration
            ms_each = [round(e / steps_done * 1e3, 3) for e in elapsed_each]
            fp8 = C["quant"] is not None
            cfg_block = {"workload": f"BASELINE.json configs[{args.config - 1}]: {C['name']}; batch {ipg} per GPU, Li={Li}+Lt={Lt} tokens, "
                                     f"{p.depth} double + {p.depth_single_blocks} single blocks, {spr} step(s) per request x {n_req} request(s) = {steps_done} "
                                     f"timed steps, repeated {R} times (median reported), hipGraph denoise loop",
                         "baseline_config": args.config, "images_per_gpu": ipg, "global_batch": world * ipg, "parallelism": f"batch-sharded replicas x{world}",
                         "nranks": nranks, "backend": backend, "calibration": calibration, "finite_output": finite,
                         "depth_override": args.depth, "lora_fuse_s": None if lora_s is None else round(lora_s, 2),
                         "row_pair_and_table_rebuild_ms_after_lora": rebuild_ms}
            result = {
                "metric": "denoise it/s, " + C["name"] + (" (image-steps per second: every loop iteration advances all images of the batch by one step)" if ipg > 1 else ""),
                "value": round(its, 4), "unit": "it/s", "n_gpus": world, "steps": steps_done, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "requests": R, "ms_per_step_each": ms_each,
                "value_range": [round(world * ipg * steps_done / max(elapsed_each), 4), round(world * ipg * steps_done / min(elapsed_each), 4)],
                "loop_its_per_gpu": round(loop_its, 4), "image_steps_per_s": round(its, 4),
                "sustained_shader_clock_ghz_each": clock_each, "board_power": power.summary() if power else None,
                "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": ("fp8_e4m3 weights x fp8_e5m2 activations (fp32 accumulate), bf16 flow" if fp8 else "bf16 (nn.Linear weights and flow)"),
                "data": "synthetic seeded request + random-init Flux weights (no checkpoint available offline)",
                "config": cfg_block,
            }
            if args.share_device:
                result.update({"share_device": True, "data": result["data"] + f"; {world} ranks with real engines SHARING cuda:0 (plumbing run of the "
                               "N-rank timed path on a one-GPU box: the value is not a scaling measurement)"})
            if dry:
                result.update({"dry_run": True, "data": "DRY RUN on a stub engine (CPU tensors): control flow only, not a measurement",
                               "amax_exchanges": model.exchanges})
            else:
                lin_flops = linear_flops_per_step(Li, Lt, batch=ipg)
                attn_flops = 4.0 * (Li + Lt) ** 2 * 128 * p.num_heads * ipg
                n_lin = sum(cnt for *_x, cnt, _e in gemm_shapes(Li, Lt))  # grouped Linear launches per step (152)
                n_attn = p.depth + p.depth_single_blocks
                src = {"achieved": None, "traffic": None, "mfma_busy": None}
                have_prof = subprocess.run(["which", "rocprofv3"], capture_output=True).returncode == 0
                # ---- the isolated probe loops of rounds 1 - 4 (back-to-back launches of one shape, warm Infinity Cache): kept as probe_* ----
                fl = by = sec = None
                gemm_table, attn_row = [], None
                if not args.no_probe:
                    fl, by, sec, gemm_table = measure_gemm_roofline(torch, ops, dev, Li, Lt, fp8=fp8, batch=ipg)
                    attn_row = measure_attention(torch, ops, dev, Li + Lt, H=p.num_heads, batch=ipg)
                # ---- in-step kernel times: one more request of this command under rocprofv3 --kernel-trace (child process) ----
                step = None
                if world == 1 and have_prof and not args.no_step_trace:
                    d = collect_step_trace(args)
                    hdr = (f"rocprofv3 --kernel-trace -- python bench.py --config {args.config} --steps {args.steps} --warmup {args.warmup} --requests 1 --no-pmc "
                           f"--no-cpu-baseline --no-step-trace --no-probe   (child of `python bench.py {' '.join(sys.argv[1:])}`; the parent's timed line: "
                           f"{ms_per_step:.3f} ms/step = {its:.2f} it/s)")
                    step = summarize_step_trace(d, header=hdr) if d else None
                    if step:
                        with open(os.path.join(d, "steady_step.txt"), "w") as f:
                            f.write(step.pop("text"))
                    if d and not args.keep_trace:  # the raw trace is ~13 MB per run; the summary is what gets committed under profiles/
                        import glob

                        for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
                            os.remove(f)
                if sec:
                    probe = {"probe_achieved": round((fl if fp8 else by) / sec / (1e12 if fp8 else 1e9), 1),
                             "probe_frac": round((fl / sec / 1e12 / FP8_PEAK_TFLOPS) if fp8 else (by / sec / 1e9 / HBM_PEAK_GBS), 4),
                             "probe_avg_launch_us": round(sec * 1e6, 2)}
                else:
                    probe = {"probe_achieved": None, "probe_frac": None, "probe_avg_launch_us": None}
                alg_bytes_step = (by * n_lin) if by else None
                if step:
                    g_s = step["gemm_ms"] * 1e-3
                    ach = (lin_flops / g_s / 1e12) if fp8 else ((alg_bytes_step / g_s / 1e9) if alg_bytes_step else None)
                    src["achieved"] = (f"in-step: rocprofv3 --kernel-trace of one extra graph-replayed request run by this command ({step['steps']} steady steps; "
                                       "summary in gpurun_out/step_trace_config%d/steady_step.txt): algorithmic work of the step's Linear launches / time of the GEMM-family kernels inside the step" % args.config)
                    avg_us = step["gemm_ms"] * 1e3 / n_lin
                else:
                    ach = probe["probe_achieved"]
                    src["achieved"] = "isolated probe loop (HIP events on the launch stream, this process): no step trace (rocprofv3 not on PATH, N > 1, or --no-step-trace)"
                    avg_us = probe["probe_avg_launch_us"]
                peak = FP8_PEAK_TFLOPS if fp8 else HBM_PEAK_GBS
                if world == 1 and fp8 and ipg == 1 and not args.no_probe and (args.pmc or (have_prof and not args.no_pmc)):
                    try:
                        collect_pmc(args.config, Li, Lt)
                        per_t, per_b = summarize_pmc(args.config, Li, Lt)
                        if per_t:
                            src["traffic"] = "live (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this command)"
                        if per_b:
                            src["mfma_busy"] = "live (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass run by this command)"
                    except Exception as ex:  # the counters must never take the measurement down
                        print(f"bench: PMC collection failed: {ex}", file=sys.stderr)
                step_pmc = None
                if world == 1 and have_prof and args.pmc_step:
                    try:
                        step_pmc = summarize_step_pmc(collect_step_pmc(args), args.config, ms_per_step=round(ms_per_step, 3))
                    except Exception as ex:  # noqa
                        print(f"bench: step PMC failed: {ex}", file=sys.stderr)
                if step_pmc is None:
                    step_pmc = read_step_pmc(args.config)
                    if step_pmc is not None:
                        step_pmc = dict(step_pmc, source="committed file profiles/" + os.path.basename(step_pmc_file(args.config)) + " (same kernel sources, earlier run)")
                else:
                    step_pmc = dict(step_pmc, source="live (bench.py --pmc-step)")
                traffic, busy = (read_pmc("traffic", args.config), read_pmc("mfma", args.config)) if ipg == 1 else (None, None)
                if traffic and src["traffic"] is None:
                    src["traffic"] = "committed file profiles/" + os.path.basename(pmc_file("traffic", args.config)) + " (same kernel sources, earlier run)"
                if busy and src["mfma_busy"] is None:
                    src["mfma_busy"] = "committed file profiles/" + os.path.basename(pmc_file("mfma", args.config)) + " (same kernel sources, earlier run)"
                busy_w = None
                if busy and gemm_table:
                    num = den = 0.0
                    for row in gemm_table:
                        if row["launch"] in busy["per_launch"]:
                            w = row["us"] * row["per_step"]
                            num, den = num + w * busy["per_launch"][row["launch"]], den + w
                    busy_w = round(num / den, 4) if den else None
                if fp8:
                    roof = {"bound": "mfma", "kernel": "gemm_ps_kernel (persistent) / gemm_w1_kernel / gemm_pp_kernel <fp8 MX-MFMA 32x32x64, 256x256 tiles> + their 128x128 tails "
                                                        f"(the {n_lin} grouped F8Linear GEMM launches of a step as the engine issues them: fused epilogues, V^T and K in the "
                                                        "attention layout included)", "unit": "TFLOP/s"}
                else:
                    roof = {"bound": "hbm", "kernel": "bf16 MFMA GEMM at M = 512 (weight-stream bound: 23.8 GB of bf16 weights per step)", "unit": "GB/s"}
                roof.update({"achieved": None if ach is None else round(ach, 1), "peak": peak, "frac": None if ach is None else round(ach / peak, 4)})
                roof.update(probe)
                roof.update({"traffic": traffic["bytes_per_launch"] if traffic else None,
                             "traffic_note": None if traffic else "no PMC values for the current kernel sources (rocprofv3 not on PATH, --no-pmc, or a batched config)",
                             "source": src,
                             "flops_per_launch": lin_flops / n_lin, "algorithmic_bytes_per_launch": by, "avg_launch_us": None if avg_us is None else round(avg_us, 2),
                             "launches_per_step": n_lin, "mfma_busy_frac_pmc": busy_w, "launches": gemm_table,
                             "step": None if not step else {k: step[k] for k in ("steps", "wall_ms", "gemm_ms", "attention_ms", "ln_ms", "other_ms", "gaps_ms", "kernel_ms", "launches")},
                             "step_pmc": None if not step_pmc else {k: v for k, v in step_pmc.items() if k not in ("how", "formula")},
                             "attention": attn_row})
                if step and attn_row is not None:
                    a_s = step["attention_ms"] * 1e-3
                    attn_row.update({"us": round(step["attention_ms"] * 1e3 / n_attn, 2), "achieved": round(attn_flops * n_attn / a_s / 1e12, 1),
                                     "frac": round(attn_flops * n_attn / a_s / 1e12 / BF16_PEAK_TFLOPS, 4), "source": "in-step (the same kernel trace)"})
                elif attn_row is not None:
                    attn_row.update({"us": attn_row["probe_us"], "achieved": attn_row["probe_achieved"], "frac": attn_row["probe_frac"], "source": "isolated probe loop"})
                result.update({
                    "reference_h100_compiled_its": H100_COMPILED.get(args.config),
                    "vs_h100_compiled": round(its / world / H100_COMPILED[args.config], 3) if args.config in H100_COMPILED else None,
                    "fp8_mfma_fraction_whole_step": round(lin_flops / (ms_per_step * 1e-3) / (FP8_PEAK_TFLOPS * 1e12), 4) if fp8 else None,
                    "ms_per_step_hipevent": round(ms_ev_v / n_ev_v, 3) if n_ev_v else None,
                    "setup_s": round(setup_s, 1),
                    "roofline": roof,
                })
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not dry:
            try:
                result["cpu_baseline"] = cpu_baseline(args.config)
            except Exception as ex:  # the baseline must never take the measurement down
                result["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {ex}"}
        else:
            result["cpu_baseline"] = None
    if world > 1 or group1:
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        emit(json.dumps(result))


def util_schedule(num_steps, image_seq_len, shift=True):
    """get_schedule + time_shift (reference flux_pipeline.py:314-344), host floats."""
    import math

    import torch

    ts = torch.linspace(1, 0, num_steps + 1)
    if shift:
        m = (1.15 - 0.5) / (4096 - 256)
        mu = m * image_seq_len + (0.5 - m * 256)
        ts = math.exp(mu) / (math.exp(mu) + (1 / ts - 1) ** 1.0)
    return ts.tolist()


if __name__ == "__main__":
    main()
