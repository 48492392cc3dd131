"""Parity at Flux-dev's REAL geometry (hidden 3072, 24 heads, mlp 12288; L = 4608 / 2816; 19 + 38 blocks), teacher-forced.

Three cases (oracle/full_geometry.py) mirror BASELINE.json configs[1] / configs[2] and the full depth.  For each:
  1. the oracle is re-run HERE on the host cores and checked against the committed samples of the run that
     oracle/gen_golden_full.py made beside the UNMODIFIED reference (bit-equal there: 2 predictions, every F8Linear output,
     every block output) -- so the tensors the engine is compared with are the reference's;
  2. the engine gets the oracle's frozen input scales and runs the same call end to end (gate iv of SURVEY.md §8c);
  3. layer by layer, TEACHER-FORCED: every stage of every block is run alone on the oracle's own input to that stage
     (fluxmi_engine_run_block / fluxmi_engine_copy_buffer), so an error cannot hide behind, or be blamed on, an upstream
     difference.  Gates (SURVEY.md §8c i-iii): quantised fp8 bytes >= 99 % identical and never more than 1 fp8 ulp apart
     (LayerNorm + modulate chains: >= 99.9 %), GEMM outputs <= 1 bf16 ulp of the oracle's `torch._scaled_mm` and of an fp64
     evaluation on sampled rows (production auto-dispatch: grouped txt+img launches, the 256x256 ping-pong / one-wave-per-SIMD
     kernels, the hybrid 128x128 peel), block outputs rel-L2 <= 1e-2.
Round 3 adds the remaining BASELINE.json configs at real geometry (oracle/full_geometry.py): Flux-schnell bf16 at the full 19 + 38 depth
(configs[0]), a batch of two through one engine at L = 4608 (configs[3]), a rank-16 LoRA fused into the fp8 weights at the 3072-wide
layer shapes incl. the uneven-rank fused qkv (configs[4]), a 4-step graph-replayed frozen Euler loop against the oracle's loop, and an
isolated LastLayer stage check.  Every case has a hidden-256 twin (`tiny_*`) that runs the same test code in a second.
Measured values are printed; profiles/r0*_parity_*.log keep them per round.
"""
import ctypes as C
import math
import os
import time

import pytest
import torch
from safetensors.torch import load_file

import flux_oracle as fo
import full_geometry as fg
from parity_util import assert_close_mag, f8_ulp_diff, round_fp64_to_bf16, ulp_diff

pytestmark = pytest.mark.gpu
def _tuning(**knobs):
    from fluxmi import _lib

    return _lib.tuning(**knobs)


def a8_min(H):
    """quantised attention output: required fraction of e5m2 bytes identical to the oracle's.  Measured 0.993-0.996 at the real geometry
    (hidden 3072; gate 0.988 = 3 x the spread seen across the pool's boxes below the worst measurement) and 0.987-0.991 on the hidden-256
    twins (few elements, L = 96: gate 0.980); the residue is e5m2 re-gridding of 1-ulp bf16 differences (profiles/r03_attention_parity_f16k.txt);
    the oracle's own CPU results move with the host"""
    return 0.988 if H >= 3072 else 0.980
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------------------------------
def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class Checks:
    """collects every gate so that one GPU run reports all measurements instead of stopping at the first miss"""

    def __init__(self, name):
        self.name, self.rows, self.fail = name, [], []

    def f8(self, what, got, ref, min_exact, max_rel):
        """quantised e5m2 bytes: fraction bit-identical + relative L2 distance of the decoded values (an ordinal 'ulp distance' is
        meaningless around zero: +2^-16 and -2^-16 are 130 codes apart)"""
        g, r = got.reshape(-1).view(torch.uint8), ref.reshape(-1).view(torch.uint8)
        exact = (g == r).float().mean().item()
        gd, rd = g.view(torch.float8_e5m2).float(), r.view(torch.float8_e5m2).float()
        rel = ((gd - rd).norm() / rd.norm().clamp_min(1e-30)).item()
        ok = exact >= min_exact and rel <= max_rel
        self.rows.append(f"  {'ok ' if ok else 'BAD'} {what:58s} fp8 bytes identical {exact:.5f} (>= {min_exact}), decoded rel-L2 {rel:.2e} (<= {max_rel:g})")
        if not ok:
            self.fail.append(what)

    def bf16(self, what, got, ref, min_exact, min_within1, max_rel):
        """bf16 tensors: fraction bit-identical, fraction within 1 bf16 ulp, relative L2 distance"""
        d = ulp_diff(got.reshape(-1), ref.reshape(-1))
        exact, within = (d == 0).float().mean().item(), (d <= 1).float().mean().item()
        rel = rel_l2(got, ref)
        ok = exact >= min_exact and within >= min_within1 and rel <= max_rel
        self.rows.append(f"  {'ok ' if ok else 'BAD'} {what:58s} bf16 identical {exact:.5f} (>= {min_exact}), <= 1 ulp {within:.5f} (>= {min_within1}), "
                         f"rel-L2 {rel:.2e} (<= {max_rel:g})")
        if not ok:
            self.fail.append(what)

    def l2(self, what, got, ref, tol):
        e = rel_l2(got, ref)
        ok = e <= tol and math.isfinite(e)
        self.rows.append(f"  {'ok ' if ok else 'BAD'} {what:58s} rel-L2 {e:.3e} (<= {tol:g})")
        if not ok:
            self.fail.append(what)
        return e

    def done(self):
        print(f"\n[{self.name}]")
        print("\n".join(self.rows))
        assert not self.fail, f"{self.name}: {len(self.fail)} gate(s) missed: {self.fail[:8]}"


def build_engine_model(case, p, sd, dev):
    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8

    cfg = util.load_config(util.ModelVersion.flux_schnell if case.get("schnell") else util.ModelVersion.flux_dev, flow_dtype="bfloat16")
    cfg.params.depth, cfg.params.depth_single_blocks = p.depth, p.depth_single_blocks
    for k, v in case.get("params", {}).items():
        setattr(cfg.params, k, v)
    model = util.load_flow_model(cfg, {k: v for k, v in sd.items()})
    model.to(dev)
    q = case["quant"]
    if q is not None:  # None: the bf16 flow (nn.Linear everywhere), the engine then runs its unfused bf16 path
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=q["modulation"], quantize_flow_embedder_layers=q["embedders"])
    return model


@torch.inference_mode()
def adopt_frozen_scales(model, orc, lora_names=()):
    """weights: must already be bit-identical (layers a LoRA was fused into on both sides are compared by the caller instead); input
    scales: taken from the oracle (== the reference's), marked frozen"""
    n = 0
    for name, st in orc.lin.items():
        if not isinstance(st, fo.F8LinearState):
            continue
        m = model.get_submodule(name)
        if name not in lora_names:
            assert m.scale.item() == st.scale.item(), f"{name}: weight scale {m.scale.item()} vs {st.scale.item()}"
            assert torch.equal(m.float8_data.cpu().view(torch.uint8), st.float8_data.view(torch.uint8)), f"{name}: float8_data"
        m._ensure_state(m.float8_data.device)
        m.input_scale.fill_(st.input_scale.item())
        m.input_scale_reciprocal.fill_(st.input_scale_reciprocal.item())
        assert m.input_scale.item() == st.input_scale.item() and m.input_scale_reciprocal.item() == st.input_scale_reciprocal.item()
        m.trial_index, m.input_scale_initialized = m.num_scale_trials, True
        n += 1
    return n


class Eng:
    """thin handle on the engine's test hooks"""

    def __init__(self, model):
        from fluxmi import _lib, ops

        self.m, self.lib, self.ops = model, _lib, ops

    def put(self, name, t, offset=0):
        t = t.contiguous()
        self.lib.call("fluxmi_engine_copy_buffer", self.m._engine, name.encode(), offset, self.ops._p(t), t.numel() * t.element_size(), 1,
                      self.ops._stream())

    def get(self, name, shape, dtype, offset=0):
        t = torch.empty(shape, dtype=dtype, device="cuda")
        self.lib.call("fluxmi_engine_copy_buffer", self.m._engine, name.encode(), offset, self.ops._p(t), t.numel() * t.element_size(), 0,
                      self.ops._stream())
        torch.cuda.synchronize()
        return t.cpu()

    def run(self, kind, idx, s0, s1, mode=1):
        self.lib.call("fluxmi_engine_run_block", self.m._engine, kind, idx, mode, s0, s1, self.ops._stream())


def sampled_fp64_gemm(ck, what, got_rows, x8, st, rows):
    """GEMM on identical fp8 operands vs an order-independent fp64 evaluation, on sampled rows (all N columns)."""
    a = x8[rows]
    ref64 = fo.scaled_mm_fp64(a, st.float8_data, st.input_scale_reciprocal, st.scale_reciprocal, st.bias)
    S = (a.double().abs() @ st.float8_data.double().abs().T) * float(st.input_scale_reciprocal * st.scale_reciprocal)
    noise = 16.0 * math.sqrt(max(a.shape[1], 256)) * 2.0 ** -24 * S * 2.0 ** 7
    try:
        ex = assert_close_mag(got_rows, round_fp64_to_bf16(ref64), mag=noise, ulps=1.05, min_exact=0.975, what=what)
        ck.rows.append(f"  ok  {what:58s} vs fp64 on {len(rows)} rows: <= 1 bf16 ulp, bit-exact {ex:.5f}")
    except AssertionError as e:
        ck.rows.append(f"  BAD {what:58s} {e}")
        ck.fail.append(what)


# ---------------------------------------------------------------------------------------------------------------------
def prepare_case(name, dev):
    import oracle_prefetch

    # host side (checkpoint + the oracle's calibrating and frozen calls): from the background worker that started on it at session start,
    # else computed here -- the same function either way
    pre = oracle_prefetch.take(name)
    case, p, sd, inp, orc, o0, o1, tr = pre if pre is not None else oracle_prefetch.compute(name, lambda m: print(m, flush=True))
    model = build_engine_model(case, p, sd, dev)
    fg.add_lora_weight_entries(tr, orc, p, case)
    # 1. the oracle on THIS host vs the run that was pinned to the reference (build container: bit-identical, see
    # profiles/r02_gen_golden_full.log).  Same code, another CPU: torch picks other GEMM / SDPA blockings (AMX vs AVX-512 bf16), the
    # per-tensor amax -- hence every input scale -- moves in its last bits, and every activation is then re-quantised on another e5m2
    # grid: the two runs differ like two fp8 runs of the reference on two hosts do (several %).  Sanity gate + report, not a parity gate.
    fixture = os.path.join(GOLDEN, f"g10_full_{name}.safetensors")
    if os.path.exists(fixture):
        want = load_file(fixture)
        dict.__setitem__(tr, "pred_calib", o0); dict.__setitem__(tr, "pred_frozen", o1)
        got = fg.digest(tr)
        n, eq, dist = fg.compare_digest(got, {k: v for k, v in want.items() if k not in ("input_scales", "weight_scales")})
        names = sorted(k for k, m in orc.lin.items() if isinstance(m, fo.F8LinearState))
        sc = torch.tensor([orc.lin[k].input_scale.item() for k in names], dtype=torch.float32)
        sc_dev = float(((sc - want["input_scales"]).abs() / want["input_scales"]).max()) if names else 0.0
        late = max(dist.items(), key=lambda kv: kv[1]) if dist else ("", 0.0)
        print(f"[{name}] oracle on this host vs the run pinned to the reference: {eq}/{n} tensors bit-identical (samples + whole-tensor "
              f"checksums); worst sample rel-L2 {late[1]:.2e} ({late[0]}); calibrating prediction {dist.get('pred_calib', 0.0):.2e}, frozen "
              f"{dist.get('pred_frozen', 0.0):.2e}; input scales: max relative deviation {sc_dev:.2e}", flush=True)
        if names:
            assert torch.equal(torch.tensor([orc.lin[k].scale.item() for k in names]), want["weight_scales"]), "weight scales differ from the pinned run"
        assert late[1] <= 0.25 and (not names or sc_dev <= 0.25), f"oracle far from the pinned reference run: {late}, scales {sc_dev:.2f}"
    lora_names = ()
    if case.get("lora"):
        lora_names = fuse_and_check_lora(name, case, p, model, orc)
    n_f8 = adopt_frozen_scales(model, orc, lora_names)
    if n_f8:
        print(f"[{name}] engine: {n_f8} F8Linear with bit-identical fp8 weights, input scales adopted from the oracle", flush=True)
    return case, p, inp, model, orc, o1, tr


@torch.inference_mode()
def fuse_and_check_lora(name, case, p, model, orc):
    """Flux.load_lora (dict form, scale 1.0) on the engine model: fluxmi_lora_fuse_f8 per layer (dequantise, fp32 B@A incl. the uneven-rank
    chunk sum of the fused qkv layers, bf16 rounding, fresh amax / scale, re-quantise) against the oracle's fuse == the reference's
    (lora_loading.py:509-577,615-631,678-689; pinned by the w8:* entries of the fixture).  The delta is an fp32 GEMM whose summation order
    differs from torch.mm's on the CPU, so a few elements land on the other side of a bf16 rounding boundary: gate = weight scale
    bit-identical AND >= 99.9 % of the fp8 bytes identical, per layer."""
    lora = fg.make_lora(p, **case["lora"])
    names = sorted({k.split(".lora_")[0] for k in lora})
    model.load_lora({k: v.clone() for k, v in lora.items()}, 1.0, name=name)
    torch.cuda.synchronize()
    worst, rows = 1.0, []
    for nm in names:
        m, st = model.get_submodule(nm), orc.lin[nm]
        same = (m.float8_data.cpu().view(torch.uint8) == st.float8_data.view(torch.uint8)).float().mean().item()
        worst = min(worst, same)
        assert m.scale.item() == st.scale.item(), f"{nm}: weight scale after the fuse {m.scale.item()} vs the oracle's {st.scale.item()}"
        assert same >= 0.999, f"{nm} {tuple(st.float8_data.shape)}: only {same:.5f} of the fused fp8 bytes match the oracle"
        rows.append(f"{nm} {tuple(st.float8_data.shape)} {same:.6f}")
    print(f"[{name}] LoRA rank {case['lora']['rank']} fused into {len(names)} F8Linear (fused-qkv layers in the uneven-rank form): weight scales "
          f"bit-identical, fp8 bytes identical >= {worst:.6f} per layer\n    " + "\n    ".join(rows[:6]) + "\n    ...", flush=True)
    return set(names)


def end_to_end(ck, name, model, inp, o1, dev, tol):
    d = {k: v.to(dev) for k, v in inp.items()}
    args = fg.call_args(d, fg.T_FROZEN)
    args = tuple(a.to(dev) for a in args)
    pred = model(*args, mode=1)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    e = ck.l2("Flux.forward end to end (fused, frozen scales) vs oracle", pred, o1, tol)
    pred2 = model(*args, mode=2)
    ck.l2("  fused (mode 1) vs unfused-frozen (mode 2) on the GPU", pred, pred2.cpu(), 2e-3)
    return e


def teacher_forced_double(ck, E, orc, tr, i, H, Lt, L, prev_img, prev_txt):
    pre = f"double_blocks.{i}"
    cat = lambda a, b: torch.cat((tr[a].reshape(Lt, -1), tr[b].reshape(L - Lt, -1)), 0)
    x_in = torch.cat((prev_txt[0], prev_img[0]), 0).cuda()                    # [L, H] txt rows first
    mods = torch.cat((tr[pre + ".img_mod.lin.out"][0], tr[pre + ".txt_mod.lin.out"][0])).cuda()
    E.put("mod", mods, offset=i * 12 * H * 2)
    Hm = 4 * H
    # stage 0: LN + modulate + quantise
    E.put("x", x_in); E.run(0, i, 0, 0)
    ck.f8(f"{pre} LN+modulate -> qkv input", E.get("a8", (L, H), torch.uint8), cat(pre + ".txt_attn.qkv.x8", pre + ".img_attn.qkv.x8").view(torch.uint8), 0.9995, 2e-3)
    # stages 1-3 on the oracle's quantised input: qkv GEMM (grouped txt+img), K relayout, attention -> quantised proj input
    E.put("a8", cat(pre + ".txt_attn.qkv.x8", pre + ".img_attn.qkv.x8").view(torch.uint8).cuda()); E.run(0, i, 1, 3)
    attn8_got = E.get("attn8", (L, H), torch.uint8)  # the default path: K (QKNorm + RoPE) and V^T straight from the GEMM epilogue
    with _tuning(fuse_kv=1):  # the GEMM's own k columns are only visible when K goes through the qkv buffer (relayout kernel; same K bits)
        E.run(0, i, 1, 1)
    qkv = E.get("qkv", (L, 3 * H), torch.bfloat16)
    ref_qkv = cat(pre + ".txt_attn.qkv.out", pre + ".img_attn.qkv.out")
    ck.bf16(f"{pre} qkv GEMM (q,k columns; V leaves as V^T)", qkv[:, :2 * H], ref_qkv[:, :2 * H], 0.980, 0.997, 2e-3)
    rows = torch.arange(Lt, L, max(1, (L - Lt) // 48))[:48]
    sampled_fp64_gemm(ck, f"{pre} img qkv GEMM", qkv[rows][:, :2 * H], tr[pre + ".img_attn.qkv.x8"], _q2(orc.lin[pre + ".img_attn.qkv"], 2 * H), rows - Lt)
    ck.f8(f"{pre} attention -> proj input", attn8_got, cat(pre + ".txt_attn.proj.x8", pre + ".img_attn.proj.x8").view(torch.uint8), a8_min(H), 2e-2)
    # stage 4 on the oracle's attention output: proj + gate*y + x
    E.put("attn8", cat(pre + ".txt_attn.proj.x8", pre + ".img_attn.proj.x8").view(torch.uint8).cuda()); E.put("x", x_in); E.run(0, i, 4, 4)
    mid = torch.cat((tr[pre + ".txt_mid"][0], tr[pre + ".img_mid"][0]), 0)
    ck.bf16(f"{pre} proj + gate*y + x", E.get("x", (L, H), torch.bfloat16), mid, 0.995, 0.998, 1e-3)
    # stage 5
    E.put("x", mid.cuda()); E.run(0, i, 5, 5)
    ck.f8(f"{pre} LN+modulate -> mlp.0 input", E.get("a8", (L, H), torch.uint8), cat(pre + ".txt_mlp.0.x8", pre + ".img_mlp.0.x8").view(torch.uint8), 0.9995, 2e-3)
    # stage 6: mlp.0 + GELU + quantise (table-driven epilogue, hybrid 256/128 tile split)
    E.put("a8", cat(pre + ".txt_mlp.0.x8", pre + ".img_mlp.0.x8").view(torch.uint8).cuda()); E.run(0, i, 6, 6)
    ck.f8(f"{pre} mlp.0 GEMM + GELU -> mlp.2 input", E.get("h8", (L, Hm), torch.uint8), cat(pre + ".txt_mlp.2.x8", pre + ".img_mlp.2.x8").view(torch.uint8), 0.998, 5e-3)
    # stage 7: mlp.2 (K = 12288) + gate*y + x
    E.put("h8", cat(pre + ".txt_mlp.2.x8", pre + ".img_mlp.2.x8").view(torch.uint8).cuda()); E.put("x", mid.cuda()); E.run(0, i, 7, 7)
    out = torch.cat((tr[pre + ".txt_out"][0], tr[pre + ".img_out"][0]), 0)
    ck.bf16(f"{pre} mlp.2 + gate*y + x", E.get("x", (L, H), torch.bfloat16), out, 0.99, 0.997, 1e-3)
    # the whole block on the oracle's input
    E.put("x", x_in); E.run(0, i, 0, 7)
    ck.l2(f"{pre} whole block, teacher-forced input", E.get("x", (L, H), torch.bfloat16), out, block_tolerance(ck, orc, tr, i, prev_img, prev_txt, out))
    return tr[pre + ".img_out"], tr[pre + ".txt_out"]


def block_tolerance(ck, orc, tr, i, prev_img, prev_txt, out):
    """Gate for a whole DoubleStreamBlock on the oracle's input: SURVEY.md 8c(iii) asks for rel-L2 <= 1e-2, but a double block chains
    three e5m2 re-quantisations behind the attention, and e5m2 turns a 1-ulp bf16 difference into a 12-25 % step on ~1 % of the
    elements.  The noise floor is measured, not assumed: the ORACLE's own block re-evaluated with an equally valid attention (exact
    fp64 softmax rounded once instead of torch's SDPA) moves by `noise`; the engine may be 1.75 x that (never tighter than 1e-2;
    measured 0.87-1.31 x over the cases below)."""
    with torch.inference_mode():
        ai, at = orc.double_block(i, prev_img, prev_txt, tr["vec"], tr["pe"], attn_fn=fo.attention_exact)
    noise = rel_l2(torch.cat((at[0], ai[0]), 0), out)
    ck.rows.append(f"  --  double_blocks.{i}: the oracle itself moves by rel-L2 {noise:.3e} when its SDPA is replaced by an exact softmax")
    return max(1e-2, 1.75 * noise)


class _q2:
    """view of an F8LinearState restricted to its first n output rows (q|k columns of a fused qkv weight)"""

    def __init__(self, st, n):
        self.float8_data, self.bias = st.float8_data[:n], None if st.bias is None else st.bias[:n]
        self.input_scale_reciprocal, self.scale_reciprocal = st.input_scale_reciprocal, st.scale_reciprocal


def teacher_forced_single(ck, E, orc, tr, i, depth, H, L, x_prev):
    pre = f"single_blocks.{i}"
    Hm, HC = 4 * H, 5 * H
    x_in = x_prev[0].cuda()
    E.put("mod", tr[pre + ".modulation.lin.out"][0].cuda(), offset=(depth * 12 * H + i * 3 * H) * 2)
    E.put("x", x_in); E.run(1, i, 0, 0)
    x8 = tr[pre + ".linear1.x8"]
    ck.f8(f"{pre} LN+modulate -> linear1 input", E.get("a8", (L, H), torch.uint8), x8.view(torch.uint8), 0.9995, 2e-3)
    # stages 1-3: linear1 (N = 21504, split epilogue), K relayout, attention
    E.put("a8", x8.view(torch.uint8).cuda()); E.run(1, i, 1, 3)
    cat8 = E.get("cat8", (L, HC), torch.uint8)  # the default path (fused K / V^T)
    with _tuning(fuse_kv=1):  # k columns through the qkv buffer for the GEMM check (see the double block)
        E.run(1, i, 1, 1)
    lin1 = tr[pre + ".linear1.out"]
    qkv = E.get("qkv", (L, 3 * H), torch.bfloat16)
    ck.bf16(f"{pre} linear1 GEMM (q,k columns)", qkv[:, :2 * H], lin1[:, :2 * H], 0.980, 0.997, 2e-3)
    rows = torch.arange(0, L, max(1, L // 48))[:48]
    sampled_fp64_gemm(ck, f"{pre} linear1 GEMM", qkv[rows][:, :2 * H], x8, _q2(orc.lin[pre + ".linear1"], 2 * H), rows)
    ref_cat8 = tr[pre + ".linear2.x8"].view(torch.uint8)
    ck.f8(f"{pre} linear1 GEMM + GELU -> linear2 input (mlp part)", cat8[:, H:], ref_cat8[:, H:], 0.998, 5e-3)
    ck.f8(f"{pre} attention -> linear2 input (attn part)", cat8[:, :H], ref_cat8[:, :H], a8_min(H), 2e-2)
    # stage 4: linear2 (K = 15360) + gate*y + x
    E.put("cat8", ref_cat8.cuda()); E.put("x", x_in); E.run(1, i, 4, 4)
    out = tr[pre + ".out"][0]
    got = E.get("x", (L, H), torch.bfloat16)
    ck.bf16(f"{pre} linear2 + gate*y + x", got, out, 0.993, 0.998, 1e-3)
    E.put("x", x_in); E.run(1, i, 0, 4)
    ck.l2(f"{pre} whole block, teacher-forced input", E.get("x", (L, H), torch.bfloat16), out, 1e-2)
    return tr[pre + ".out"]


def teacher_forced_last_layer(ck, E, orc, tr, p, H, Lt, L, x_final, o1):
    """LastLayer.forward alone (flux_model.py:499-503) on the oracle's final residual stream and the oracle's adaLN vectors: stage 0
    (1 + scale) * LayerNorm(x) + shift, stage 1 the bf16 Linear 3072 -> 64 (never fp8, float8_quantize.py:476)."""
    import torch.nn.functional as F

    with torch.inference_mode():
        mod = orc.lin["final_layer.adaLN_modulation.1"](F.silu(tr["vec"]))       # [B, 2H] = shift | scale
        shift, scale = mod.chunk(2, dim=1)
        x_img = x_final[:, Lt:, :]
        ref_fin = (1 + scale[:, None, :]) * fo.layer_norm(x_img) + shift[:, None, :]
    E.put("mod", mod[0].contiguous().cuda(), offset=(p.depth * 12 * H + p.depth_single_blocks * 3 * H) * 2)
    E.put("x", x_final[0].contiguous().cuda())
    E.run(2, 0, 0, 0, mode=2)
    ck.bf16("final_layer LN + modulate (stage 0)", E.get("fin", (L - Lt, H), torch.bfloat16), ref_fin[0], 0.995, 0.9995, 2e-3)
    E.put("fin", ref_fin[0].contiguous().cuda())
    E.run(2, 0, 1, 1, mode=2)
    ck.bf16("final_layer.linear (stage 1, bf16 GEMM N = 64)", E.get("pred_s", (L - Lt, o1.shape[-1]), torch.bfloat16), o1[0], 0.98, 0.997, 2e-3)
    E.put("x", x_final[0].contiguous().cuda())
    E.run(2, 0, 0, 1, mode=2)
    ck.bf16("final_layer whole, teacher-forced input", E.get("pred_s", (L - Lt, o1.shape[-1]), torch.bfloat16), o1[0], 0.97, 0.995, 3e-3)


@pytest.mark.parametrize("name", ["tiny_2p2_L96", "c2_2p2_L4608", "c3_2p2_L2816", "c2_default_1p1_L3392", "c2_ragged_1p1_L3257"])
def test_teacher_forced_blocks_at_real_geometry(dev, name):
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    H = p.hidden_size
    Lt = case["txt_len"]
    L = Lt + (case["height"] // 16) * (case["width"] // 16)
    e = end_to_end(ck, name, model, inp, o1, dev, 7e-2)
    # gate (iv): the engine is no further from the reference's bf16 flow than the reference's own fp8 path (x 1.25)
    orc_bf16 = fo.FluxOracle({k: v for k, v in orc.sd.items()}, p, quantize=None)
    with torch.inference_mode():
        rb = orc_bf16.forward(*fg.call_args(inp, fg.T_FROZEN))
    d = {k: v.to(dev) for k, v in inp.items()}
    pred = model(*tuple(a.to(dev) for a in fg.call_args(d, fg.T_FROZEN)), mode=1)
    d_ref, d_got = rel_l2(o1, rb), rel_l2(pred, rb)
    ok = d_got <= 1.25 * d_ref
    ck.rows.append(f"  {'ok ' if ok else 'BAD'} {'gate (iv): distance to the bf16 flow path':58s} engine {d_got:.3e} vs oracle-fp8 {d_ref:.3e} (x1.25)")
    if not ok:
        ck.fail.append("gate iv")
    E = Eng(model)
    img, txt = tr["img_in.out"], tr["txt_in.out"]
    for i in range(p.depth):
        img, txt = teacher_forced_double(ck, E, orc, tr, i, H, Lt, L, img, txt)
    x = torch.cat((txt, img), 1)
    for i in range(p.depth_single_blocks):
        x = teacher_forced_single(ck, E, orc, tr, i, p.depth, H, L, x)
    teacher_forced_last_layer(ck, E, orc, tr, p, H, Lt, L, x, o1)
    ck.done()


def test_full_depth_19_38(dev):
    """The whole Flux-dev depth (19 double + 38 single blocks, hidden 3072): (1) every one of the 57 blocks run alone on the ORACLE's
    input to that block (teacher-forced; the engine's own modulation vectors) must reproduce the oracle's output to rel-L2 <= 1e-2;
    (2) end to end the engine must be no further from the reference's bf16 flow than the reference's own fp8 path is, x 1.25
    (SURVEY.md 8c gate iv) -- the fp8 model is chaotic over 57 residual blocks (the oracle itself moves by ~1e-1 between two host
    CPUs, see the message printed by prepare_case), so a fixed end-to-end distance to the fp8 oracle would gate on noise;
    (3) the free-running drift after every block is reported."""
    name = "c2_19p38_L320"
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    H, Lt = p.hidden_size, case["txt_len"]
    L = Lt + (case["height"] // 16) * (case["width"] // 16)
    d = {k: v.to(dev) for k, v in inp.items()}
    args = tuple(a.to(dev) for a in fg.call_args(d, fg.T_FROZEN))
    pred = model(*args, mode=1)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    e2e = rel_l2(pred, o1)
    ck.l2("  fused (mode 1) vs unfused-frozen (mode 2) on the GPU", pred, model(*args, mode=2).cpu(), 2e-2)
    orc_bf16 = fo.FluxOracle({k: v for k, v in orc.sd.items()}, p, quantize=None)
    with torch.inference_mode():
        rb = orc_bf16.forward(*fg.call_args(inp, fg.T_FROZEN))
    d_ref, d_got = rel_l2(o1, rb), rel_l2(pred, rb)
    ok = d_got <= 1.25 * d_ref
    ck.rows.append(f"  {'ok ' if ok else 'BAD'} {'gate (iv): distance to the bf16 flow path, 57 blocks':58s} engine {d_got:.3e} vs oracle-fp8 {d_ref:.3e} (x1.25); "
                   f"engine vs oracle-fp8 {e2e:.3e}")
    if not ok:
        ck.fail.append("gate iv")
    # (1) teacher-forced, block by block (the unfused forward above left the engine's own modulation vectors of this call in `mod`)
    E = Eng(model)
    tf = []
    prev = torch.cat((tr["txt_in.out"][0], tr["img_in.out"][0]), 0)
    tol, noise_d = [], []
    p_img, p_txt = tr["img_in.out"], tr["txt_in.out"]
    for i in range(p.depth):
        E.put("x", prev.cuda()); E.run(0, i, 0, 7)
        ref = torch.cat((tr[f"double_blocks.{i}.txt_out"][0], tr[f"double_blocks.{i}.img_out"][0]), 0)
        tf.append(rel_l2(E.get("x", (L, H), torch.bfloat16), ref))
        with torch.inference_mode():  # noise floor of this block: the oracle with an exact softmax instead of SDPA (see block_tolerance)
            ai, at = orc.double_block(i, p_img, p_txt, tr["vec"], tr["pe"], attn_fn=fo.attention_exact)
        noise_d.append(rel_l2(torch.cat((at[0], ai[0]), 0), ref))
        tol.append(max(1e-2, 1.75 * noise_d[-1]))
        p_img, p_txt = tr[f"double_blocks.{i}.img_out"], tr[f"double_blocks.{i}.txt_out"]
        prev = ref
    for i in range(p.depth_single_blocks):
        E.put("x", prev.cuda()); E.run(1, i, 0, 4)
        ref = tr[f"single_blocks.{i}.out"][0]
        tf.append(rel_l2(E.get("x", (L, H), torch.bfloat16), ref))
        tol.append(1e-2)
        prev = ref
    ok = all(math.isfinite(v) and v <= t for v, t in zip(tf, tol))
    nd = p.depth
    ck.rows.append(f"  {'ok ' if ok else 'BAD'} {'each of the 57 blocks on the oracle input (teacher-forced)':58s} double blocks: worst rel-L2 "
                   f"{max(tf[:nd]):.3e} (gate per block = max(1e-2, 1.75 x the oracle's own SDPA-vs-exact-softmax movement, worst {max(noise_d):.3e}; worst engine / noise "
                   f"ratio {max(a / b for a, b in zip(tf[:nd], noise_d)):.2f})); "
                   f"single blocks: worst {max(tf[nd:]):.3e} (<= 1e-2); median of all {sorted(tf)[len(tf) // 2]:.3e}")
    if not ok:
        ck.fail.append("teacher-forced blocks")
    # (3) free-running drift (report)
    E.put("x", torch.cat((tr["txt_in.out"][0], tr["img_in.out"][0]), 0).cuda())
    drift = []
    for i in range(p.depth):
        E.run(0, i, 0, 7)
        drift.append(rel_l2(E.get("x", (L, H), torch.bfloat16), torch.cat((tr[f"double_blocks.{i}.txt_out"][0], tr[f"double_blocks.{i}.img_out"][0]), 0)))
    for i in range(p.depth_single_blocks):
        E.run(1, i, 0, 4)
        drift.append(rel_l2(E.get("x", (L, H), torch.bfloat16), tr[f"single_blocks.{i}.out"][0]))
    ck.rows.append("  --  free-running residual stream vs the oracle after blocks 1, 10, 19 (double) / 20, 38, 57: "
                   + ", ".join(f"{drift[k]:.2e}" for k in (0, 9, 18, 19, 37, 56)))
    assert all(math.isfinite(v) for v in drift)
    # (4) round 6: the stated PER-PIXEL tolerance at the full depth.  One Euler jump from t = T_FROZEN to 0 with each of the three predictions of this
    # call (engine fp8 | oracle fp8 | oracle bf16 = the reference's bf16 flow path), unpack, decode with the real FLUX VAE geometry (engine latents:
    # the native VAE; oracle latents: the oracle VAE under autocast), into_bytes' uint8 map: the engine's pixels must be no further from the
    # reference-bf16 image than the reference's own fp8 image is, x 1.25 (mean |delta|) / - 1.94 dB (PSNR)   [tests/pixel_parity.py, flux_pipeline.py:619-663]
    import pixel_parity as pp
    import vae_oracle as vo
    from modules.autoencoder import AutoEncoder, AutoEncoderParams

    ae = AutoEncoder(AutoEncoderParams(**vo.FULL_PARAMS))
    ae_sd = vo.synth_state_dict({k: v.shape for k, v in ae.state_dict().items()}, seed=7)
    ae.load_state_dict(ae_sd, strict=True)
    ae.to(dev)
    Hh, Ww = case["height"], case["width"]
    x0 = lambda pr: fo.unpack_latent((inp["img"].float() - fg.T_FROZEN * pr.float().cpu()).to(torch.bfloat16).float(), Hh, Ww)
    with torch.inference_mode():
        px_e = pp.to_uint8(ae.decode(x0(pred).to(dev)))
        px_o8 = pp.to_uint8(vo.decode(ae_sd, vo.FULL_PARAMS, x0(o1), autocast=True))
        px_o16 = pp.to_uint8(vo.decode(ae_sd, vo.FULL_PARAMS, x0(rb), autocast=True))
    yard, m_e = pp.pixel_metrics(px_o8, px_o16), pp.pixel_metrics(px_e, px_o16)
    ok = m_e["mean_abs"] <= 1.25 * yard["mean_abs"] and m_e["psnr_db"] >= yard["psnr_db"] - 1.94
    ck.rows.append(f"  {'ok ' if ok else 'BAD'} {'per-pixel tolerance after 57 blocks (one Euler jump, real VAE)':58s} engine vs reference-bf16: {pp.fmt(m_e)}; "
                   f"yardstick reference-fp8 vs reference-bf16: {pp.fmt(yard)}")
    if not ok:
        ck.fail.append("per-pixel tolerance at full depth")
    ck.done()


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the remaining BASELINE.json configs at real geometry
# ---------------------------------------------------------------------------------------------------------------------
def _x_in(tr, key_txt, key_img, b):
    return torch.cat((tr[key_txt][b], tr[key_img][b]), 0)


def teacher_forced_all_blocks(ck, name, model, orc, tr, p, case, mode, single_tol=1e-2, double_tol=None):
    """every block alone on the ORACLE's input to it (whole blocks, every batch element; the engine's own modulation vectors of the
    preceding forward call are in `mod`).  Double blocks of fp8 models are gated against the oracle's measured SDPA-vs-exact-softmax noise
    (block_tolerance), everything else against a fixed rel-L2."""
    E = Eng(model)
    H, Lt = p.hidden_size, case["txt_len"]
    L = Lt + (case["height"] // 16) * (case["width"] // 16)
    B = tr["img_in.out"].shape[0]
    cat_b = lambda kt, ki: torch.cat([_x_in(tr, kt, ki, b) for b in range(B)], 0)
    prev = cat_b("txt_in.out", "img_in.out")
    p_img, p_txt = tr["img_in.out"], tr["txt_in.out"]
    tf, tol, noise_d = [], [], []
    for i in range(p.depth):
        E.put("x", prev.cuda()); E.run(0, i, 0, 7, mode=mode)
        ref = cat_b(f"double_blocks.{i}.txt_out", f"double_blocks.{i}.img_out")
        tf.append(rel_l2(E.get("x", (B * L, H), torch.bfloat16), ref))
        if double_tol is None:
            with torch.inference_mode():
                ai, at = orc.double_block(i, p_img, p_txt, tr["vec"], tr["pe"], attn_fn=fo.attention_exact)
            noise_d.append(rel_l2(torch.cat([torch.cat((at[b], ai[b]), 0) for b in range(B)], 0), ref))
            tol.append(max(1e-2, 1.75 * noise_d[-1]))
        else:
            tol.append(double_tol)
        p_img, p_txt = tr[f"double_blocks.{i}.img_out"], tr[f"double_blocks.{i}.txt_out"]
        prev = ref
    for i in range(p.depth_single_blocks):
        E.put("x", prev.cuda()); E.run(1, i, 0, 4, mode=mode)
        ref = torch.cat([tr[f"single_blocks.{i}.out"][b] for b in range(B)], 0)
        tf.append(rel_l2(E.get("x", (B * L, H), torch.bfloat16), ref))
        tol.append(single_tol)
        prev = ref
    ok = all(math.isfinite(v) and v <= t for v, t in zip(tf, tol))
    nd = p.depth
    ck.rows.append(f"  {'ok ' if ok else 'BAD'} {f'each of the {len(tf)} blocks on the oracle input (teacher-forced, B = {B})':58s} double blocks: worst rel-L2 "
                   f"{max(tf[:nd]):.3e} (gate " + (f"{double_tol:g}" if double_tol is not None else f"max(1e-2, 1.75 x oracle noise, worst {max(noise_d):.3e})")
                   + f"); single blocks: worst {max(tf[nd:]):.3e} (<= {single_tol:g}); median of all {sorted(tf)[len(tf) // 2]:.3e}")
    if not ok:
        ck.fail.append("teacher-forced blocks")
    return E, prev


@pytest.mark.parametrize("name", ["tiny_schnell_bf16_L48", "c1_schnell_bf16_1p1_L4352", "c1_schnell_bf16_19p38_L512"])
def test_schnell_bf16_flow_at_full_depth(dev, name):
    """BASELINE.json configs[0] at its REAL geometry: Flux-schnell (no guidance embedder), hidden 3072, all 19 + 38 blocks, 256x256 + 256 text
    tokens, bf16 nn.Linear everywhere -- the engine's bf16 path (bf16 MFMA GEMMs at M = 256 / 512, unfused sequencing).  Gates: every block alone
    on the oracle's input rel-L2 <= 7e-3 double / 5e-3 single (bf16 rounding + fp32 summation order only: no fp8 anywhere; measured 3.5e-3 /
    1.1e-3), the whole 57-block forward <= 3.5e-2: measured 1.8e-2, which is what the ORACLE itself moves between two host CPUs (1.77e-2
    between the build container and the GPU box, printed by prepare_case) -- rounding-level differences amplified through 57 residual blocks."""
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    assert not model.f8_modules() and orc.n_f8() == 0 and not p.guidance_embed
    d = {k: v.to(dev) for k, v in inp.items()}
    args = tuple(a.to(dev) for a in fg.call_args(d, fg.T_FROZEN))
    pred = model(*args[:6], None)  # schnell: no guidance
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    ck.l2(f"Flux.forward end to end, bf16 flow, {p.depth}+{p.depth_single_blocks} blocks vs oracle", pred, o1, 3.5e-2)
    E, x_final = teacher_forced_all_blocks(ck, name, model, orc, tr, p, case, mode=2, single_tol=5e-3, double_tol=7e-3)
    # the bf16 flow's denoise loop as bench.py --config 1 runs it: hipGraph replay == eager launches, bit for bit.  At hidden 3072 the M = 512
    # launches with K >= 6144 run split-K, whose partial-tile scratch must belong to the ENGINE (the graph is captured on a private stream and
    # replayed on the caller's; a scratch keyed by stream failed under capture in round 4)
    ts = [1.0, 0.75, 0.5, 0.25, 0.0]
    lat_g = model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], ts, guidance=fg.GUIDANCE, use_graph=True)
    lat_e = model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], ts, guidance=fg.GUIDANCE, use_graph=False)
    torch.cuda.synchronize()
    same = torch.isfinite(lat_g).all().item() and torch.equal(lat_g.view(torch.int16), lat_e.view(torch.int16))
    ck.rows.append(f"  {'ok ' if same else 'BAD'} 4-step bf16 denoise loop: graph replay == eager, bit for bit")
    if not same:
        ck.fail.append("bf16 graph vs eager")
    ck.done()
    if p.depth >= 19:
        fg.drop_sd_cache()  # 24 GB of synthetic checkpoint shared with test_full_depth_19_38


@pytest.mark.parametrize("name", ["tiny_B2_L96", "c4_B2_1p1_L4608", "c4_B2_ragged_1p1_L3257"])
def test_batch_of_two_at_real_geometry(dev, name):
    """BASELINE.json configs[3] on one GPU: TWO different samples through one engine at L = 4608 (batch strides of every buffer, grouped GEMM
    launches with 2 x (txt, img) groups, attention over B x heads) against the oracle's / reference's batch-2 call; and each sample of the
    batch bit-identical to the same sample run alone (samples never interact: flux_model.py has no cross-sample op)."""
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    d = {k: v.to(dev) for k, v in inp.items()}
    args = tuple(a.to(dev) for a in fg.call_args(d, fg.T_FROZEN))
    pred = model(*args, mode=1)
    torch.cuda.synchronize()
    assert pred.shape[0] == 2 and torch.isfinite(pred).all()
    for b in range(2):
        ck.l2(f"Flux.forward (fused, B = 2) sample {b} vs oracle", pred[b], o1[b], 7e-2)
    teacher_forced_all_blocks(ck, name, model, orc, tr, p, case, mode=1)
    for b in range(2):
        one = model(*tuple(a[b:b + 1].contiguous() for a in args), mode=1)
        same = torch.equal(one[0].view(torch.int16), pred[b].view(torch.int16))
        ck.rows.append(f"  {'ok ' if same else 'BAD'} sample {b} alone (B = 1) == the same sample inside the batch of 2, bit for bit")
        if not same:
            ck.fail.append(f"batch independence {b}")
    ck.done()


@pytest.mark.parametrize("name", ["tiny_lora_L96", "c5_lora_2p2_L4608"])
def test_lora_fused_at_real_geometry(dev, name):
    """BASELINE.json configs[4]: rank-16 LoRA fused into the calibrated fp8 model at the real layer shapes (3072x3072, 9216x3072 uneven-rank
    fused qkv, 12288x3072, 3072x12288, 21504x3072, 3072x15360).  prepare_case: weight bytes / scales of all 26 fused layers against the
    oracle == reference (fuse_and_check_lora); here: the forward through the fused weights."""
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    d = {k: v.to(dev) for k, v in inp.items()}
    args = tuple(a.to(dev) for a in fg.call_args(d, fg.T_FROZEN))
    pred = model(*args, mode=1)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    ck.l2("Flux.forward through the LoRA-fused fp8 weights vs oracle", pred, o1, 7e-2)
    ck.l2("  fused (mode 1) vs unfused-frozen (mode 2) on the GPU", pred, model(*args, mode=2).cpu(), 2e-3)
    teacher_forced_all_blocks(ck, name, model, orc, tr, p, case, mode=1)
    ck.done()


@pytest.mark.parametrize("name", ["tiny_loop4_L96", "c2_loop4_2p2_L4608"])
def test_frozen_denoise_loop_at_real_geometry(dev, name):
    """The hot path's outer loop at real geometry: 4 frozen Euler steps, hipGraph-replayed (engine_denoise: step-ahead modulation table, one
    captured step graph, device-side step counter) against the oracle's loop == the reference's (flux_pipeline.py:619-651, pinned in the
    fixture).  The per-step prediction is within 4-5e-2 of the oracle's fp8 prediction (e5m2 re-gridding noise, see the teacher-forced test);
    over the loop the latents stay within 5e-2.  Also: graph replay == eager launches, bit for bit."""
    case, p, inp, model, orc, o1, tr = prepare_case(name, dev)
    ck = Checks(name)
    ts = fg.loop_schedule(case)
    d = {k: v.to(dev) for k, v in inp.items()}
    assert model.calibration_state()[0]
    lat = model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], ts, guidance=fg.GUIDANCE, use_graph=True)
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all()
    ref = tr["loop_latents"]
    ck.l2(f"latents after {case['loop_steps']} frozen Euler steps (hipGraph replay) vs the oracle's loop", lat, ref, 5e-2)
    moved = rel_l2(ref, inp["img"])
    ck.rows.append(f"  --  (the loop moves the latents by rel-L2 {moved:.3f} from the initial noise)")
    lat2 = model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], ts, guidance=fg.GUIDANCE, use_graph=False)
    same = torch.equal(lat.view(torch.int16), lat2.view(torch.int16))
    ck.rows.append(f"  {'ok ' if same else 'BAD'} graph-replayed loop == eager loop, bit for bit")
    if not same:
        ck.fail.append("graph vs eager")
    # kernel-selection knobs compute the same bits at real geometry too: K by the relayout kernel (fuse_kv 1, 0) instead of the GEMM epilogue
    # (the fused-K tiles sum their squares in the relayout kernel's order), one-tile-per-workgroup GEMMs (their fused-K epilogue sums in
    # another order, within 1 ulp: tests/test_ops_gpu.py -- hence with the relayout kernel), no weight prefetch
    from fluxmi import _lib

    for knobs in (dict(fuse_kv=1), dict(fuse_kv=0), dict(gemm_persist=0, fuse_kv=1), dict(prefetch=0), dict(w_pairs=0)):
        with _lib.tuning(**knobs):
            lat3 = model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], ts, guidance=fg.GUIDANCE, use_graph=True)
        same = torch.equal(lat.view(torch.int16), lat3.view(torch.int16))
        ck.rows.append(f"  {'ok ' if same else 'BAD'} latents under tuning {knobs} == default, bit for bit" + ("" if same else f" (rel-L2 {rel_l2(lat3, lat):.3e})"))
        if not same:
            ck.fail.append(f"tuning {knobs}")
    ck.done()
