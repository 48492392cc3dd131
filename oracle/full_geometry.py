"""Full-geometry parity cases (Flux-dev widths: hidden 3072 = 24 heads x 128, mlp 12288)  --  TEST INFRASTRUCTURE.

Shared by oracle/gen_golden_full.py (build container: runs the UNMODIFIED reference beside the oracle, asserts equality, writes
tests/golden/g10_full_*.safetensors) and tests/test_full_geometry_gpu.py (GPU box: re-runs the oracle on the host cores, checks
it against the committed reference samples, then checks the HIP engine against the oracle layer by layer, teacher-forced).

Cases (one per BASELINE.json config, plus the full depth and the loop):
  c2_2p2_L4608   2 double + 2 single blocks, 1024x1024 (Li 4096) + Lt 512, quantize_modulation, embedders in bf16       configs[1]
  c3_2p2_L2816   2 + 2 blocks, 768x768 (Li 2304) + Lt 512, quantize_modulation + quantize_flow_embedder_layers           configs[2]
  c2_19p38_L320  the whole 19 + 38-block model, 256x256 (Li 256) + Lt 64 (fp8 error accumulation through 57 residual blocks)
  c1_schnell_bf16_19p38_L512  Flux-schnell (no guidance embedder), ALL 19 + 38 blocks, 256x256 (Li 256) + Lt 256, bf16 nn.Linear everywhere
                 (no fp8): configs[0] at its real geometry and depth; the checkpoint is c2_19p38_L320's minus guidance_in
  c4_B2_1p1_L4608  1 + 1 blocks of c2_2p2_L4608's model on a batch of TWO different samples in one call (batch strides at L = 4608)  configs[3]
  c5_lora_2p2_L4608  c2_2p2_L4608's model with a rank-16 LoRA on every attention / MLP linear (fused-qkv layers in the reference's "uneven
                 rank" form) fused into the fp8 weights at scale 1.0 AFTER calibration (lora_loading.py:509-577,635-693)  configs[4]
  c2_loop4_2p2_L4608  c2_2p2_L4608's model: one calibrating call, then a 4-step frozen Euler loop (flux_pipeline.py:619-651)
  c2_default_1p1_L3392  1 + 1 blocks at the resolution FluxPipeline.generate defaults to (width 720, height 1024, flux_pipeline.py:526-540):
                 Li = 64 x 45 = 2880, L = 3392 = 13.25 row tiles of 256 (53 key tiles of 64)
  c2_ragged_1p1_L3257  1 + 1 blocks at 720 x 976: Li = 61 x 45 = 2745, L = 3257 (ODD: ragged last GEMM row tile of 185 rows, ragged last query
                 block and key tile in attention, padded V^T rows), 13 x 3 = 39 attention tasks per XCD = a thin last round (the balanced grid)
  c4_B2_ragged_1p1_L3257  the same odd L with a batch of TWO different samples (batch strides over ragged rows, 2 x (txt, img) GEMM groups each
                 with its own ragged last tile, attention over B x heads with a ragged last block)
  c1_schnell_bf16_1p1_L4352  Flux-schnell's bf16 flow (no fp8) at 1024x1024 + Lt 256: the bf16 MFMA GEMMs at LARGE M (17 row tiles), which the
                 256x256 config-1 case (M = 512) never launches
Protocol per case: call 1 calibrates (every F8Linear takes its first amax trial, float8_quantize.py:220-238), the input scales are
then frozen (`input_scale_initialized = True`, what the reference does after its 13th call, :239-246) and call 2 runs frozen with
every intermediate recorded.  A full tensor at L = 4608 is tens of MB, so fixtures hold SAMPLES (first 256 + 768 evenly strided
elements) plus an order-independent checksum of every recorded tensor.
"""
from __future__ import annotations

import time
from typing import Dict

import torch

import flux_oracle as fo

CASES = {
    "c2_2p2_L4608": dict(depth=2, single=2, height=1024, width=1024, txt_len=512, quant=dict(modulation=True, embedders=False),
                         w_seed=11, in_seed=21, trace="all"),
    "c3_2p2_L2816": dict(depth=2, single=2, height=768, width=768, txt_len=512, quant=dict(modulation=True, embedders=True),
                         w_seed=12, in_seed=22, trace="all"),
    "c2_19p38_L320": dict(depth=19, single=38, height=256, width=256, txt_len=64, quant=dict(modulation=True, embedders=False),
                          w_seed=13, in_seed=23, trace="blocks"),
    "c1_schnell_bf16_19p38_L512": dict(depth=19, single=38, height=256, width=256, txt_len=256, quant=None, w_seed=13, in_seed=25, trace="blocks",
                                       params=dict(guidance_embed=False), schnell=True, sd_from="c2_19p38_L320"),
    "c4_B2_1p1_L4608": dict(depth=1, single=1, height=1024, width=1024, txt_len=512, quant=dict(modulation=True, embedders=False),
                            w_seed=11, in_seed=26, trace="blocks", batch=2),
    "c5_lora_2p2_L4608": dict(depth=2, single=2, height=1024, width=1024, txt_len=512, quant=dict(modulation=True, embedders=False),
                              w_seed=11, in_seed=21, trace="blocks", lora=dict(rank=16, seed=5)),
    "c2_loop4_2p2_L4608": dict(depth=2, single=2, height=1024, width=1024, txt_len=512, quant=dict(modulation=True, embedders=False),
                               w_seed=11, in_seed=21, trace="none", loop_steps=4),
    "c2_default_1p1_L3392": dict(depth=1, single=1, height=1024, width=720, txt_len=512, quant=dict(modulation=True, embedders=False),
                                 w_seed=16, in_seed=29, trace="all"),
    "c2_ragged_1p1_L3257": dict(depth=1, single=1, height=976, width=720, txt_len=512, quant=dict(modulation=True, embedders=False),
                                w_seed=16, in_seed=30, trace="all"),
    "c4_B2_ragged_1p1_L3257": dict(depth=1, single=1, height=976, width=720, txt_len=512, quant=dict(modulation=True, embedders=False),
                                   w_seed=16, in_seed=31, trace="blocks", batch=2),
    "c1_schnell_bf16_1p1_L4352": dict(depth=1, single=1, height=1024, width=1024, txt_len=256, quant=None, w_seed=17, in_seed=32, trace="blocks",
                                      params=dict(guidance_embed=False), schnell=True),
    # harness checks only (no fixture): the same test code on models that run in a second
    "tiny_schnell_bf16_L48": dict(depth=2, single=2, height=64, width=64, txt_len=32, quant=None, w_seed=15, in_seed=27, trace="blocks", schnell=True,
                                  params=dict(hidden_size=256, num_heads=2, context_in_dim=128, vec_in_dim=64, guidance_embed=False)),
    "tiny_B2_L96": dict(depth=2, single=2, height=128, width=128, txt_len=32, quant=dict(modulation=True, embedders=False), w_seed=14, in_seed=28,
                        trace="blocks", batch=2, params=dict(hidden_size=256, num_heads=2, context_in_dim=128, vec_in_dim=64)),
    "tiny_lora_L96": dict(depth=2, single=2, height=128, width=128, txt_len=32, quant=dict(modulation=True, embedders=False), w_seed=14, in_seed=24,
                          trace="blocks", lora=dict(rank=4, seed=6), params=dict(hidden_size=256, num_heads=2, context_in_dim=128, vec_in_dim=64)),
    "tiny_loop4_L96": dict(depth=2, single=2, height=128, width=128, txt_len=32, quant=dict(modulation=True, embedders=False), w_seed=14, in_seed=24,
                           trace="none", loop_steps=4, params=dict(hidden_size=256, num_heads=2, context_in_dim=128, vec_in_dim=64)),
    "tiny_2p2_L96": dict(depth=2, single=2, height=128, width=128, txt_len=32, quant=dict(modulation=True, embedders=False),
                         w_seed=14, in_seed=24, trace="all",
                         params=dict(hidden_size=256, num_heads=2, context_in_dim=128, vec_in_dim=64)),
}
T_CALIB, T_FROZEN, GUIDANCE = 1.0, 0.75, 3.5
N_HEAD, N_STRIDED = 256, 768


def params_for(case: dict) -> fo.FluxParams:
    return fo.FluxParams(depth=case["depth"], depth_single_blocks=case["single"], **case.get("params", {}))


_SD_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}  # full-depth checkpoints take minutes to synthesise: shared between cases of one process


def drop_sd_cache():
    _SD_CACHE.clear()


def make_case(name: str, synth):
    """-> (case dict, FluxParams, state dict (CPU bf16), inputs dict).  `synth` = the fluxmi.synth module.  Nobody may modify the
    returned tensors in place (cases share checkpoints)."""
    case = CASES[name]
    p = params_for(case)
    if case.get("sd_from"):
        base = CASES[case["sd_from"]]
        bp = params_for(base)
        assert (bp.depth, bp.depth_single_blocks, bp.hidden_size) == (p.depth, p.depth_single_blocks, p.hidden_size) and base["w_seed"] == case["w_seed"]
        key = (case["sd_from"], base["w_seed"])
        if key not in _SD_CACHE:
            _SD_CACHE[key] = synth.make_state_dict(bp, seed=base["w_seed"])
        sd = {k: v for k, v in _SD_CACHE[key].items() if p.guidance_embed or not k.startswith("guidance_in.")}
    else:
        key = (name if p.depth >= 19 else None, case["w_seed"])
        if key[0] is not None:
            if key not in _SD_CACHE:
                _SD_CACHE[key] = synth.make_state_dict(p, seed=case["w_seed"])
            sd = dict(_SD_CACHE[key])
        else:
            sd = synth.make_state_dict(p, seed=case["w_seed"])
    inp = synth.make_inputs(p, case["height"], case["width"], case["txt_len"], batch=case.get("batch", 1), seed=case["in_seed"], real_tokens=32)
    return case, p, sd, inp


def make_lora(p, rank: int, seed: int) -> Dict[str, torch.Tensor]:
    """rank-`rank` LoRA on every attention / MLP linear (BFL-dotted keys = what Flux.load_lora takes as a dict, lora_loading.py:608-612):
    the fused qkv layers get the reference's "uneven rank" form A [3r, K], B [3N', r] (lora_loading.py:533-541)."""
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


def add_lora_weight_entries(tr: dict, orc, p, case) -> None:
    """LoRA cases: the fused + re-quantised fp8 weights themselves become part of the digest ("w8:<layer>")."""
    if not case.get("lora"):
        return
    keys = make_lora(p, **case["lora"])
    for nm, st in orc.lin.items():
        if isinstance(st, fo.F8LinearState) and (nm + ".lora_A.weight") in keys:
            dict.__setitem__(tr, "w8:" + nm, st.float8_data)


def loop_schedule(case) -> list:
    Li = (case["height"] // 16) * (case["width"] // 16)
    return fo.get_schedule(case["loop_steps"], Li, shift=not case.get("schnell", False))


def call_args(inp, t: float):
    tv = torch.full((inp["img"].shape[0],), t, dtype=torch.bfloat16)
    gv = torch.full((inp["img"].shape[0],), GUIDANCE, dtype=torch.bfloat16)
    return (inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], tv, inp["y"], gv)


class FilteredTrace(dict):
    """A trace dict that drops what the case does not record (the 19+38 case keeps block outputs only: a full trace is ~30 GB)."""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode

    def __setitem__(self, k, v):
        if self.mode == "none":
            return
        if self.mode == "all" or k in ("vec", "pe", "img_in.out", "txt_in.out") or k.endswith((".img_out", ".txt_out")) or (
                k.startswith("single_blocks") and k.endswith(".out") and k.count(".") == 2):
            super().__setitem__(k, v)


def run_oracle(name: str, p, sd, inp, log=print):
    """calibrating call, freeze, [fuse the case's LoRA], traced frozen call, [the case's frozen Euler loop].
    -> (oracle, pred_calib, pred_frozen, trace); the loop's final latents are trace["loop_latents"]."""
    case = CASES[name]
    t0 = time.time()
    orc = fo.FluxOracle(sd, p, quantize=case["quant"])
    log(f"[{name}] oracle built ({orc.n_f8()} F8Linear) in {time.time() - t0:.1f} s")
    t0 = time.time()
    with torch.inference_mode():
        pred0 = orc.forward(*call_args(inp, T_CALIB))
        log(f"[{name}] oracle calibrating call {time.time() - t0:.1f} s")
        if case["quant"] is not None:
            orc.freeze_input_scales()
        if case.get("lora"):
            t0 = time.time()
            orc.fuse_lora(make_lora(p, **case["lora"]), 1.0)  # no input-scale recalibration after fusing, as in the reference
            log(f"[{name}] oracle LoRA fuse {time.time() - t0:.1f} s")
        tr = FilteredTrace(case["trace"])
        t0 = time.time()
        pred1 = orc.forward(*call_args(inp, T_FROZEN), trace=tr)
        log(f"[{name}] oracle frozen call {time.time() - t0:.1f} s, {len(tr)} tensors recorded")
        if case.get("loop_steps"):
            t0 = time.time()
            lat = fo.denoise(orc, inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], loop_schedule(case), guidance=GUIDANCE)
            dict.__setitem__(tr, "loop_latents", lat)
            log(f"[{name}] oracle {case['loop_steps']}-step frozen Euler loop {time.time() - t0:.1f} s")
    return orc, pred0, pred1, tr


def sample_index(n: int) -> torch.Tensor:
    if n <= N_HEAD + N_STRIDED:
        return torch.arange(n)
    head = torch.arange(N_HEAD)
    strided = N_HEAD + (torch.arange(N_STRIDED, dtype=torch.int64) * (n - 1 - N_HEAD)) // (N_STRIDED - 1)
    return torch.cat([head, strided])


def as_words(t: torch.Tensor) -> torch.Tensor:
    """raw storage words of a bf16 / fp8 / fp32 tensor as integers (for samples and checksums)."""
    t = t.contiguous()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.uint8)
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16)
    if t.dtype == torch.float32:
        return t.view(torch.int32)
    return t


def digest(trace: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """name -> samples (storage words) and name + '#sum' -> [sum of words (int64), sum of |value| (fp64 bits)]."""
    out = {}
    for k, v in trace.items():
        if not torch.is_tensor(v):
            continue
        w = as_words(v).reshape(-1)
        out[k] = w[sample_index(w.numel())].clone()
        out[k + "#sum"] = torch.stack([w.to(torch.int64).sum(), v.double().abs().sum().view(torch.int64)])
    return out


def _decode(w: torch.Tensor) -> torch.Tensor:
    if w.dtype == torch.int16:
        return w.view(torch.bfloat16).double()
    if w.dtype == torch.uint8:
        return w.view(torch.float8_e5m2).double()  # every quantised activation on this path is e5m2 (float8_quantize.py:43)
    if w.dtype == torch.int32:
        return w.view(torch.float32).double()
    return w.double()


def compare_digest(got: Dict[str, torch.Tensor], want: Dict[str, torch.Tensor]):
    """-> (n_tensors, n bit-identical (samples AND whole-tensor checksums), {name: relative L2 distance of the samples}).
    Bit-identity is what the build container shows (same host as the pinned run).  Another host CPU selects other GEMM / SDPA
    blockings inside torch (AMX vs AVX-512 bf16 paths): rounding-level differences that the e5m2 re-quantisation and 57 residual
    blocks amplify, so callers gate tightly on the EARLY tensors only and report the rest.  Quantised (uint8) tensors are skipped
    when the scales differ: their bytes are then not comparable."""
    n = eq = 0
    dist = {}
    for k, w in want.items():
        if k.endswith("#sum"):
            continue
        n += 1
        g = got[k]
        if torch.equal(got[k + "#sum"], want[k + "#sum"]) and torch.equal(g, w):
            eq += 1
            dist[k] = 0.0
            continue
        if w.dtype == torch.uint8:
            continue
        gd, wd = _decode(g), _decode(w)
        fin = torch.isfinite(gd) & torch.isfinite(wd)
        dist[k] = float((gd[fin] - wd[fin]).norm() / wd[fin].norm().clamp_min(1e-30))
    return n, eq, dist
