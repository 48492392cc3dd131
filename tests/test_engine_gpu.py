"""GPU parity of the whole path: Flux.forward / the denoise loop (native engine) vs the CPU oracle.

Small models (hidden 256 = 2 heads x 128, 2+2 blocks) so the oracle finishes in seconds; the reference's own
module tree and state-dict layout; configs mirror BASELINE.json's list:
  (1) bf16 flow, no fp8            (2) fp8, quantize_modulation=True      (3) + quantize_flow_embedder_layers=True
  (4) batch > 1                    (5) LoRA fused into the fp8 weights
Tolerances: the fused kernels re-apply the reference's bf16 rounding points, so everything up to reduction
order is bit-identical; residual differences come from attention (flash vs math softmax) and fp32 summation order.
  * bf16 model: pred vs oracle rel-L2 <= 1e-2 per call (measured 5e-3)
  * fp8 models: e5m2 activations have 2 mantissa bits, so a 1-bf16-ulp upstream difference flips ~1/64 of the quantised
    bytes by 25 %: swapping F.scaled_dot_product_attention for an exact fp64 softmax INSIDE the oracle already moves its
    own output by 3.5e-2 rel-L2 (measured).  Gates: rel-L2(engine, oracle-fp8) <= 6e-2 and, SURVEY.md §8c
    gate (iv), rel-L2(engine, oracle-bf16) <= 1.25 x rel-L2(oracle-fp8, oracle-bf16)
  * calibrated input scales (max over 12 running amax values of chaotic activations): within 30 % of the oracle's,
    >= 30 % bit-identical; weight scales and float8_data bytes: bit-identical
  * fused (mode 1) vs unfused-frozen (mode 2) on the GPU: rel-L2 <= 2e-3  (same scales, same rounding points)
  * hipGraph denoise loop vs per-step forward + Euler on the GPU: bit-identical
"""
import copy
import os
import queue
import time

import pytest
import torch

import flux_oracle as fo

pytestmark = pytest.mark.gpu


def tiny_config(schnell=False, **kw):
    import util

    cfg = util.load_config(util.ModelVersion.flux_schnell if schnell else util.ModelVersion.flux_dev, flow_dtype="bfloat16", **kw)
    p = cfg.params
    p.hidden_size, p.num_heads, p.depth, p.depth_single_blocks, p.context_in_dim, p.vec_in_dim = 256, 2, 2, 2, 128, 64
    return cfg


def build(cfg, quant, dev, seed=0):
    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8
    from fluxmi import synth

    sd = synth.make_state_dict(cfg.params, seed=seed)
    model = util.load_flow_model(cfg, {k: v.clone() for k, v in sd.items()})
    model.to(dev)
    if quant is not None:
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=quant["modulation"], quantize_flow_embedder_layers=quant["embedders"])
    oracle = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, fo.FluxParams(**cfg.params.model_dump()), quantize=quant)
    return model, oracle, sd


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def to_dev(inp, dev):
    return {k: v.to(dev) for k, v in inp.items()}


QUANTS = {
    "bf16": None,
    "fp8": dict(modulation=True, embedders=False),
    "fp8_emb": dict(modulation=True, embedders=True),
    "fp8_nomod": dict(modulation=False, embedders=False),
}


@pytest.mark.parametrize("qname", list(QUANTS))
@pytest.mark.parametrize("shape", [(64, 64, 32, 2), (48, 80, 40, 1)])
def test_forward_matches_oracle_through_calibration(dev, qname, shape):
    """15 consecutive Flux.forward calls (12 calibration + freeze + 2 frozen/fused) track the oracle call by call."""
    from fluxmi import synth

    H, W, Lt, B = shape
    cfg = tiny_config()
    model, oracle, sd = build(cfg, QUANTS[qname], dev)
    oracle_bf16 = fo.FluxOracle({k: v.clone() for k, v in sd.items()}, fo.FluxParams(**cfg.params.model_dump()), quantize=None)
    assert len(model.f8_modules()) == oracle.n_f8()
    inp = synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=3, real_tokens=8)
    dinp = to_dev(inp, dev)
    worst = 0.0
    for step in range(15):
        t = torch.full((B,), 1.0 - 0.06 * step, dtype=torch.bfloat16)
        g = torch.full((B,), 3.5, dtype=torch.bfloat16)
        ref = oracle.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], g)
        got = model(dinp["img"], dinp["img_ids"], dinp["txt"], dinp["txt_ids"], t.to(dev), dinp["y"], g.to(dev))
        assert torch.isfinite(got).all()
        e = rel_l2(got, ref)
        worst = max(worst, e)
        if QUANTS[qname] is None:
            assert e <= 1e-2, f"{qname} call {step}: rel-L2 {e:.3e}"
        else:
            assert e <= 6e-2, f"{qname} call {step}: rel-L2 vs fp8 oracle {e:.3e}"
            if step in (0, 7, 14):
                rb = oracle_bf16.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], g)
                d_ref, d_got = rel_l2(ref, rb), rel_l2(got, rb)
                assert d_got <= 1.25 * d_ref, f"{qname} call {step}: vs bf16 flow {d_got:.3e} > 1.25 x {d_ref:.3e}"
    if QUANTS[qname] is not None:
        frozen, _ = model.calibration_state()
        assert frozen
        names = [n for n, m in oracle.lin.items() if isinstance(m, fo.F8LinearState)]
        exact = 0
        for n in names:
            mod = model.get_submodule(n)
            so, sg = oracle.lin[n].input_scale.item(), mod.input_scale.item()
            assert abs(sg - so) <= 0.30 * so, f"{n}: input_scale {sg} vs oracle {so}"
            exact += int(sg == so)
            assert mod.scale.item() == oracle.lin[n].scale.item(), f"{n}: weight scale"
            assert torch.equal(mod.float8_data.cpu().view(torch.uint8), oracle.lin[n].float8_data.view(torch.uint8)), f"{n}: float8_data"
        print(f"[{qname}] {exact}/{len(names)} calibrated input scales bit-identical to the oracle's")
        assert exact >= 0.3 * len(names), f"only {exact}/{len(names)} input scales bit-identical"
    print(f"[{qname} {shape}] worst rel-L2 over 15 calls: {worst:.3e}")


def test_full_width_blocks_match_oracle(dev):
    """Flux-dev's real widths (hidden 3072 = 24 heads x 128, mlp 12288, single-block K = 15360) on 1 + 1 blocks: exercises the
    production tile configs (ping-pong for K = 3072, one-wave-per-SIMD for K >= 8192), the fused V^T epilogue with 24 heads, the
    raw-Q attention and the modulation GEMM at the shapes of BASELINE.json configs[1]; sequence kept short for the CPU oracle."""
    import util
    from fluxmi import synth

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16")
    p = cfg.params
    p.depth, p.depth_single_blocks = 1, 1
    quant = QUANTS["fp8"]
    model, oracle, _ = build(cfg, quant, dev, seed=1)
    H, W, Lt, B = 256, 256, 64, 1   # Li = 256, L = 320
    inp = synth.make_inputs(p, H, W, Lt, batch=B, seed=4, real_tokens=16)
    dinp = to_dev(inp, dev)
    g = torch.full((B,), 3.5, dtype=torch.bfloat16)
    worst = 0.0
    for step in range(15):
        t = torch.full((B,), 1.0 - 0.05 * step, dtype=torch.bfloat16)
        ref = oracle.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], g)
        got = model(dinp["img"], dinp["img_ids"], dinp["txt"], dinp["txt_ids"], t.to(dev), dinp["y"], g.to(dev))
        assert torch.isfinite(got).all()
        e = rel_l2(got, ref)
        worst = max(worst, e)
        assert e <= 6e-2, f"full-width call {step}: rel-L2 vs fp8 oracle {e:.3e}"
    assert model.calibration_state()[0]
    # fused (mode 1) == unfused-frozen (mode 2) at these widths too
    t = torch.full((B,), 0.3, dtype=torch.bfloat16, device=dev)
    args = (dinp["img"], dinp["img_ids"], dinp["txt"], dinp["txt_ids"], t, dinp["y"], g.to(dev))
    a, b = model(*args, mode=1), model(*args, mode=2)
    assert rel_l2(a, b) <= 2e-3
    print(f"[full width] worst rel-L2 over 15 calls: {worst:.3e}; fused vs unfused rel-L2 {rel_l2(a, b):.3e}")


def test_fused_equals_unfused_and_graph_equals_eager(dev):
    from fluxmi import synth

    cfg = tiny_config()
    model, oracle, _ = build(cfg, QUANTS["fp8"], dev)
    B, H, W, Lt = 2, 64, 64, 32
    inp = to_dev(synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=5, real_tokens=8), dev)
    ts = fo.get_schedule(16, (H // 16) * (W // 16))
    g = torch.full((B,), 3.5, dtype=torch.bfloat16, device=dev)
    # 13 calibrating calls through the denoise loop itself
    lat = model.denoise(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts[:14], guidance=3.5, use_graph=False)
    assert model.calibration_state()[0]
    t = torch.full((B,), 0.4, dtype=torch.bfloat16, device=dev)
    args = (lat, inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], g)
    fused = model(*args, mode=1)
    unfused = model(*args, mode=2)
    e = rel_l2(fused, unfused)
    frac = (fused == unfused).float().mean().item()
    print(f"fused vs unfused: rel-L2 {e:.3e}, bit-identical fraction {frac:.4f}")
    assert e <= 2e-3
    # graph replay == eager per-step loop, bit for bit
    ts2 = ts[:9]
    a = model.denoise(lat, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts2, guidance=3.5, use_graph=True)
    b = model.denoise(lat, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts2, guidance=3.5, use_graph=False)
    assert torch.equal(a, b)
    c = lat.clone()
    for t_curr, t_prev in zip(ts2[:-1], ts2[1:]):
        tv = torch.full((B,), t_curr, dtype=torch.bfloat16, device=dev)
        pred = model(c, inp["img_ids"], inp["txt"], inp["txt_ids"], tv, inp["y"], g, mode=1)
        c = c + (t_prev - t_curr) * pred
    assert torch.equal(a, c), f"graph loop vs python loop: rel-L2 {rel_l2(a, c):.3e}"
    # a second request re-uses the captured graph
    a2 = model.denoise(lat, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts2, guidance=3.5, use_graph=True)
    assert torch.equal(a, a2)
    # kernel-selection knobs are baked into a captured graph: changing the tuning struct re-captures, and none of them may change a bit --
    # the weight prefetch riding on attention / the 216-tile GEMMs (extra workgroups that only read), the persistent GEMM, the table epilogues
    from fluxmi import _lib

    for knobs in (dict(prefetch=0), dict(prefetch=2), dict(gemm_persist=0), dict(qlut=0), dict(fuse_kv=1), dict(fuse_kv=0), dict(w_pairs=0), dict(a_pairs=0)):
        with _lib.tuning(**knobs):
            a3 = model.denoise(lat, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts2, guidance=3.5, use_graph=True)
        assert torch.equal(a, a3), f"latents change under tuning {knobs}: rel-L2 {rel_l2(a3, a):.3e}"
    a4 = model.denoise(lat, inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts2, guidance=3.5, use_graph=True)
    assert torch.equal(a, a4)


@pytest.mark.parametrize("schnell", [False, True])
def test_denoise_loop_matches_oracle(dev, schnell):
    """G6: the Euler loop incl. calibration-phase steps (flux_pipeline.py:619-651) vs the oracle, fp8 and bf16."""
    from fluxmi import synth

    for qname in ("bf16", "fp8"):
        cfg = tiny_config(schnell=schnell)
        model, oracle, _ = build(cfg, QUANTS[qname], dev)
        B, H, W, Lt = 1, 64, 64, 32
        inp = synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=7, real_tokens=8)
        dinp = to_dev(inp, dev)
        n = 4 if schnell else 16
        ts = fo.get_schedule(n, (H // 16) * (W // 16), shift=not schnell)
        ref = fo.denoise(oracle, inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts, guidance=3.5)
        got = model.denoise(dinp["img"], dinp["img_ids"], dinp["txt"], dinp["txt_ids"], dinp["y"], ts, guidance=3.5)
        e = rel_l2(got, ref)
        print(f"[{qname} schnell={schnell}] latents after {n} steps: rel-L2 {e:.3e}, max abs {(got.float().cpu() - ref.float()).abs().max().item():.3e}")
        assert e <= (1e-2 if qname == "bf16" else 6e-2)


def test_lora_fuse_end_to_end(dev):
    """Config 5: rank-16 LoRA (even rank on proj/mlp, 'uneven' 3r on fused qkv) fused at scale 1.0, then unfused."""
    from fluxmi import synth

    cfg = tiny_config()
    model, oracle, _ = build(cfg, QUANTS["fp8"], dev)
    g = torch.Generator().manual_seed(21)
    lora = {}
    H = cfg.params.hidden_size
    for i in range(2):
        for s in ("img", "txt"):
            lora[f"double_blocks.{i}.{s}_attn.qkv.lora_A.weight"] = torch.randn(3 * 16, H, generator=g) * 0.05
            lora[f"double_blocks.{i}.{s}_attn.qkv.lora_B.weight"] = torch.randn(3 * H, 16, generator=g) * 0.05
            lora[f"double_blocks.{i}.{s}_attn.proj.lora_A.weight"] = torch.randn(16, H, generator=g) * 0.05
            lora[f"double_blocks.{i}.{s}_attn.proj.lora_B.weight"] = torch.randn(H, 16, generator=g) * 0.05
            lora[f"double_blocks.{i}.{s}_attn.proj.alpha"] = 8.0
        lora[f"single_blocks.{i}.linear2.lora_A.weight"] = torch.randn(16, 5 * H, generator=g) * 0.05
        lora[f"single_blocks.{i}.linear2.lora_B.weight"] = torch.randn(H, 16, generator=g) * 0.05
    before = {n: model.get_submodule(n.split(".lora")[0]).float8_data.clone() for n in lora if n.endswith("lora_A.weight")}
    model.load_lora(copy.deepcopy(lora), 1.0, name="t")
    oracle.fuse_lora(lora, 1.0)
    from parity_util import assert_f8_close

    for n in before:
        name = n.split(".lora")[0]
        mod, st = model.get_submodule(name), oracle.lin[name]
        assert abs(mod.scale.item() - st.scale.item()) <= 1e-6 * st.scale.item(), name
        assert_f8_close(mod.float8_data, st.float8_data, max_ulp=1, min_exact=0.995, what=name)
        assert not torch.equal(mod.float8_data, before[n])
    inp = synth.make_inputs(cfg.params, 64, 64, 32, batch=1, seed=9, real_tokens=8)
    dinp = to_dev(inp, dev)
    t = torch.full((1,), 0.7, dtype=torch.bfloat16)
    gv = torch.full((1,), 3.5, dtype=torch.bfloat16)
    ref = oracle.forward(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], t, inp["y"], gv)
    got = model(dinp["img"], dinp["img_ids"], dinp["txt"], dinp["txt_ids"], t.to(dev), dinp["y"], gv.to(dev))
    assert rel_l2(got, ref) <= 6e-2
    assert model.unload_lora("t")
    oracle.fuse_lora(lora, 1.0, sign=-1.0)
    for n in before:
        name = n.split(".lora")[0]
        assert_f8_close(model.get_submodule(name).float8_data, oracle.lin[name].float8_data, max_ulp=1, min_exact=0.99, what="unfuse " + name)


def test_batch_sharded_calibration_matches_whole_batch(dev):
    """SURVEY.md 8e-3 / float8_quantize.py:227: the reference takes amax over the WHOLE batch, so a batch of 2 sharded over two replicas
    must calibrate exactly like a batch of 2 on one GPU.  Two engines (one sample each) run the 13 calibrating + 3 frozen steps
    concurrently from two host threads with the per-layer amax exchange installed (Flux.enable_amax_exchange; here the reduction is a
    rendezvous between the two threads, over RCCL it is all_reduce(MAX)) -> every input scale and the latents are BIT-identical to
    the single-engine batch-2 run.  Without the exchange the scales differ (checked)."""
    import threading

    from fluxmi import synth

    cfg = tiny_config()
    B, H, W, Lt = 2, 64, 64, 32
    inp = to_dev(synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=31, real_tokens=8), dev)
    inp["img"][1] *= 1.7  # make the two samples' activation ranges differ
    inp["txt"][1] *= 2.5
    ts = fo.get_schedule(16, (H // 16) * (W // 16))
    whole, _, _ = build(cfg, QUANTS["fp8"], dev)
    ref = whole.denoise(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts, guidance=3.5)
    assert whole.calibration_state()[0]
    names = [n for n, m in whole.named_modules() if m in set(whole.f8_modules())]

    def run_pair(exchange):
        reps = [build(cfg, QUANTS["fp8"], dev)[0] for _ in range(2)]
        bar = threading.Barrier(2)
        slot = [None, None]
        outs, errs = [None, None], []

        def make_reduce(r):
            def reduce_fn(t):
                slot[r] = t
                bar.wait(timeout=120)
                m = torch.maximum(slot[0], slot[1])
                bar.wait(timeout=120)
                t.copy_(m)
            return reduce_fn

        def work(r):
            try:
                if exchange:
                    reps[r].enable_amax_exchange(make_reduce(r))
                sl = slice(r, r + 1)
                # eager loop: two engines capturing hipGraphs from two threads of ONE process trip over each other's legacy-stream
                # operations (one process per GPU in production); graph replay == eager bit for bit is checked elsewhere
                outs[r] = reps[r].denoise(inp["img"][sl], inp["img_ids"][sl], inp["txt"][sl], inp["txt_ids"][sl], inp["y"][sl], ts, guidance=3.5,
                                          use_graph=False)
                torch.cuda.synchronize()
            except Exception as e:  # noqa
                errs.append(e)
                bar.abort()

        if exchange:
            th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
            [t.start() for t in th]
            [t.join(timeout=600) for t in th]
        else:
            work(0); work(1)
        assert not errs, errs
        return reps, torch.cat(outs, 0)

    reps, lat = run_pair(True)
    for n in names:
        s_ref = whole.get_submodule(n).input_scale.item()
        for r in range(2):
            assert reps[r].get_submodule(n).input_scale.item() == s_ref, f"{n}: replica {r} scale differs from the whole-batch run"
    assert torch.equal(lat, ref), f"sharded latents differ from the whole-batch run: rel-L2 {rel_l2(lat, ref):.3e}"
    reps2, _ = run_pair(False)
    differ = sum(reps2[0].get_submodule(n).input_scale.item() != reps2[1].get_submodule(n).input_scale.item() for n in names)
    print(f"without the exchange {differ}/{len(names)} input scales differ between the replicas; with it 0 (and == the batch-2 run)")
    assert differ > 0


def _two_process_worker(rank, world, port, backend, q, release):
    """One rank of test_two_process_sharded_calibration_over_a_process_group: the REAL engine on cuda:0, one sample of the batch, the
    in-step amax exchange through torch.distributed (RCCL if it takes two ranks on one device, else gloo with the 1.2 KB amax vector
    staged through the host)."""
    import os
    import sys

    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch
        import torch.distributed as td

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for pth in (os.path.join(root, "flux-fp8-api_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
            if pth not in sys.path:
                sys.path.insert(0, pth)
        import flux_oracle as fo
        from fluxmi import dist as fdist
        from fluxmi import synth
        from test_engine_gpu import QUANTS, build, tiny_config, to_dev

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        td.init_process_group(backend=backend, rank=rank, world_size=world)
        cfg = tiny_config()
        B, H, W, Lt = 2, 64, 64, 32
        inp = to_dev(synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=31, real_tokens=8), dev)
        inp["img"][1] *= 1.7
        inp["txt"][1] *= 2.5
        # rank 0 owns the request: one flat broadcast of [txt | vec | noise] (fluxmi/dist.py), then every rank takes its shard
        txt, vec, img = inp["txt"], inp["y"], inp["img"]
        if rank != 0:
            txt, vec, img = torch.zeros_like(txt), torch.zeros_like(vec), torch.zeros_like(img)
        if backend == "gloo":
            t_, v_, i_ = fdist.broadcast_request(txt.cpu(), vec.cpu(), img.cpu(), src=0)
            txt, vec, img = t_.to(dev), v_.to(dev), i_.to(dev)
        else:
            txt, vec, img = fdist.broadcast_request(txt, vec, img, src=0)
        lo, hi = fdist.shard_bounds(B, rank, world)
        ts = fo.get_schedule(16, (H // 16) * (W // 16))
        model = build(cfg, QUANTS["fp8"], dev)[0]
        n_x = [0]
        if backend == "gloo":
            def reduce_fn(t):  # the engine hands over a device tensor on the current stream: stage the few floats through the host
                h = t.detach().cpu()
                td.all_reduce(h, op=td.ReduceOp.MAX)
                t.copy_(h)
                n_x[0] += 1
        else:
            def reduce_fn(t):
                td.all_reduce(t, op=td.ReduceOp.MAX)
                n_x[0] += 1
        model.enable_amax_exchange(reduce_fn)
        out = model.denoise(img[lo:hi], inp["img_ids"][lo:hi], txt[lo:hi], inp["txt_ids"][lo:hi], vec[lo:hi], ts, guidance=3.5)
        torch.cuda.synchronize()
        model.enable_amax_exchange(False)
        names = [n for n, m in model.named_modules() if m in set(model.f8_modules())]
        scales = {n: model.get_submodule(n).input_scale.item() for n in names}
        # after an all-reduce over every rank: the group really has `world` members (a silently degraded group would not)
        rid = torch.tensor([float(rank), 1.0])
        if backend != "gloo":
            rid = rid.to(dev)
        td.all_reduce(rid, op=td.ReduceOp.SUM)
        # plain Python objects only: a torch tensor crosses an mp.Queue by fd-passing from a thread INSIDE the sender, which races with
        # the sender's exit (the parent's q.get then raises ConnectionRefusedError).  bf16 latents travel as their int16 bit patterns.
        lat = out.detach().cpu().contiguous()
        q.put((rank, "ok", (lat.view(torch.int16).numpy().tobytes(), tuple(lat.shape)), scales, n_x[0], td.get_backend(),
               [float(v) for v in rid.cpu()]))
        td.barrier()
        td.destroy_process_group()
    except Exception as e:  # noqa
        import traceback

        q.put((rank, "error", traceback.format_exc(), None, 0, backend, None))
    # stay alive until the parent holds BOTH results (or gives up): the queue's feeder thread has then certainly flushed
    release.wait(timeout=600)


def test_two_process_sharded_calibration_over_a_process_group(dev):
    """BASELINE configs[3] (batch sharded over GPUs, one process per GPU), as far as ONE GPU can show it: two PROCESSES, each with its
    own engine on cuda:0 and one sample of a batch of two, joined by a torch.distributed process group -- RCCL (backend "nccl") if it
    accepts two ranks on one device, otherwise gloo with the per-layer amax vector staged through the host.  The 13 calibrating steps
    run the in-step amax all_reduce(MAX) from inside the engine's hook (fluxmi_engine_set_amax_exchange); the frozen scales of BOTH
    ranks and the gathered latents must equal the whole-batch run on one engine BIT FOR BIT (the reference takes amax over the whole
    batch, float8_quantize.py:227; loop: flux_pipeline.py:505-512, 619-651)."""
    import socket

    import torch.multiprocessing as mp
    from fluxmi import synth

    cfg = tiny_config()
    B, H, W, Lt = 2, 64, 64, 32
    inp = to_dev(synth.make_inputs(cfg.params, H, W, Lt, batch=B, seed=31, real_tokens=8), dev)
    inp["img"][1] *= 1.7
    inp["txt"][1] *= 2.5
    ts = fo.get_schedule(16, (H // 16) * (W // 16))
    whole = build(cfg, QUANTS["fp8"], dev)[0]
    ref = whole.denoise(inp["img"], inp["img_ids"], inp["txt"], inp["txt_ids"], inp["y"], ts, guidance=3.5).cpu()
    names = [n for n, m in whole.named_modules() if m in set(whole.f8_modules())]
    want = {n: whole.get_submodule(n).input_scale.item() for n in names}
    ctx = mp.get_context("spawn")
    tried = []
    for backend in ("nccl", "gloo"):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        q = ctx.Queue()
        release = ctx.Event()
        procs = [ctx.Process(target=_two_process_worker, args=(r, 2, port, backend, q, release)) for r in range(2)]
        for p in procs:
            p.start()
        res, why = [], None
        deadline = time.time() + (240 if backend == "nccl" else 600)
        while len(res) < len(procs) and time.time() < deadline:
            try:
                res.append(q.get(timeout=2.0))
            except queue.Empty:
                # a rank that died without reporting (RCCL refuses two ranks on one device on some stacks -> abort) will never report
                if any(p.exitcode not in (None, 0) for p in procs) and q.empty():
                    why = f"a rank exited with {[p.exitcode for p in procs]} before reporting"
                    break
            except Exception as e:  # noqa: recorded, never swallowed
                why = f"q.get raised {type(e).__name__}: {e}"
                break
        if why is None and len(res) < len(procs):
            why = f"timed out with {len(res)} of {len(procs)} results"
        release.set()
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
        ok = len(res) == 2 and all(r[1] == "ok" for r in res)
        tried.append((backend, ok, why, [r[2][-400:] if r[1] != "ok" else "ok" for r in res]))
        if ok:
            break
    assert ok, f"no backend completed the two-process run: {tried}"
    res.sort(key=lambda r: r[0])
    print(f"two-process sharded calibration over backend {res[0][5]!r} ({'RCCL, two ranks on one device' if backend == 'nccl' else 'gloo, amax staged through the host'}); "
          f"{res[0][4]} in-step amax exchanges per rank; rank-id all-reduce {res[0][6]}")
    assert res[0][6] == [1.0, 2.0] and res[1][6] == [1.0, 2.0], "the process group did not span both ranks"
    assert res[0][4] > 0 and res[0][4] == res[1][4], "both ranks must have gone through the same number of in-step exchanges"
    for r in res:
        bad = [n for n in names if r[3][n] != want[n]]
        assert not bad, f"rank {r[0]}: {len(bad)} input scales differ from the whole-batch run (e.g. {bad[:3]})"
    lat = torch.cat([torch.frombuffer(bytearray(b), dtype=torch.int16).view(torch.bfloat16).reshape(shp) for b, shp in (res[0][2], res[1][2])], 0)
    assert torch.equal(lat, ref), f"two-process latents differ from the whole-batch run: rel-L2 {rel_l2(lat, ref):.3e}"


def test_pipeline_generate_latents(dev):
    """Pipeline surface: load from a ModelSpec, calibrate via compile(), generate from embeddings (drop-in call shape)."""
    from flux_pipeline import FluxPipeline
    from fluxmi import synth

    cfg = tiny_config()
    cfg.text_enc_max_length = 32
    sd = synth.make_state_dict(cfg.params, seed=0)
    pipe = FluxPipeline.load_pipeline_from_config(cfg, state_dict=sd)
    pipe.compile()
    assert pipe.model.calibration_state()[0]
    g = torch.Generator().manual_seed(1)
    prompt = {"txt": 0.1 * torch.randn(1, 32, 128, generator=g), "vec": torch.randn(1, 64, generator=g)}
    out, seed = pipe.generate(prompt, width=64, height=96, num_steps=6, seed=123, return_seed=True, output_type="latent", silent=True)
    assert seed == 123 and out.shape == (1, 16, 12, 8) and torch.isfinite(out).all()
    out2 = pipe.generate(prompt, width=64, height=96, num_steps=6, seed=123, output_type="latent", silent=True)
    assert torch.equal(out, out2)
    with pytest.raises(ValueError):
        pipe.model(torch.zeros(1, 4, dtype=torch.bfloat16, device=dev), None, torch.zeros(1, 4, dtype=torch.bfloat16, device=dev), None, None, None)


def test_vae_decoder_matches_reference_fixture(dev):
    """AutoEncoder.decode (native, NHWC bf16, conv = im2col + MFMA GEMM) vs the unmodified reference decoder's
    outputs stored in tests/golden/g8_vae.safetensors (fp32, and under torch.autocast(bf16) on CPU).  Tolerance: the native path
    must be as close to the fp32 result as the reference's own bf16-autocast path is (x1.5)."""
    import os

    from safetensors.torch import load_file

    from modules.autoencoder import AutoEncoder, AutoEncoderParams

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_vae.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    ae = AutoEncoder(AutoEncoderParams(resolution=32, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=4,
                                       scale_factor=0.3611, shift_factor=0.1159))
    ae.load_state_dict(sd, strict=True)
    ae.to(dev)
    out = ae.decode(g["z"].to(dev)).float().cpu()
    assert out.shape == g["ref_fp32"].shape and torch.isfinite(out).all()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    e_native, e_ref = rel(out, g["ref_fp32"]), rel(g["ref_autocast"], g["ref_fp32"])
    print(f"VAE decode: native vs fp32 reference {e_native:.3e}; reference autocast vs fp32 {e_ref:.3e}; native vs oracle-autocast "
          f"{rel(out, g['oracle_autocast']):.3e}")
    assert e_native <= 1.5 * e_ref
    assert rel(out, g["oracle_autocast"]) <= 1.5e-2


def test_vae_encoder_matches_reference_fixture(dev):
    """AutoEncoder.encode_moments / encode (native; Downsample = stride-2 gather mode of im2col) vs the
    unmodified reference encoder's outputs in tests/golden/g8_vae.safetensors.  Tolerance: as close to the fp32 moments as the
    reference's own bf16-autocast path is (x1.5); the sampled latent is checked with the reference's own noise draw."""
    import os

    from safetensors.torch import load_file

    from modules.autoencoder import AutoEncoder, AutoEncoderParams

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_vae.safetensors"))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    ae = AutoEncoder(AutoEncoderParams(resolution=32, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2], num_res_blocks=1, z_channels=4,
                                       scale_factor=0.3611, shift_factor=0.1159))
    ae.load_state_dict(sd, strict=True)
    ae.to(dev)
    x = g["enc_x"].to(dev)
    mom = ae.encode_moments(x).float().cpu()
    assert mom.shape == g["enc_moments_fp32"].shape and torch.isfinite(mom).all()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    e_native, e_ref = rel(mom, g["enc_moments_fp32"]), rel(g["enc_moments_autocast"], g["enc_moments_fp32"])
    print(f"VAE encode: native vs fp32 reference {e_native:.3e}; reference autocast vs fp32 {e_ref:.3e}; native vs oracle-autocast "
          f"{rel(mom, g['enc_oracle_moments_autocast']):.3e}")
    assert e_native <= 1.5 * e_ref
    assert rel(mom, g["enc_oracle_moments_autocast"]) <= 2e-2
    z = ae.encode(x, noise=g["enc_noise"].to(dev)).float().cpu()
    assert z.shape == g["enc_encode_fp32"].shape
    assert rel(z, g["enc_encode_fp32"]) <= 1.5 * e_ref + 1e-3
    # the default path draws its own noise: same mean, different sample
    z2 = ae.encode(x).float().cpu()
    assert z2.shape == z.shape and torch.isfinite(z2).all() and not torch.equal(z2, z)
    # decoder-only checkpoints refuse to encode on random weights
    ae.encoder_loaded = False
    with pytest.raises(RuntimeError):
        ae.encode(x)


def test_vae_full_size_matches_reference_fixture(dev):
    """The REAL FLUX autoencoder geometry (ch 128, ch_mult [1,2,4,4], 2 res blocks, z 16; reference util.py:99-110) on a 256x256
    image: native decode / encode_moments vs the unmodified reference's fp32 outputs stored in tests/golden/g11_vae_full.safetensors
    (oracle/gen_golden_vae_full.py; weights rebuilt from the same seed on both sides).  Gate as at the small geometry: the native bf16
    path must be as close to the fp32 reference as the reference's own torch.autocast(bf16) run is (x1.5)."""
    import os
    import sys

    from safetensors.torch import load_file

    import vae_oracle as vo
    from modules.autoencoder import AutoEncoder, AutoEncoderParams

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_vae_full.safetensors"))
    ae = AutoEncoder(AutoEncoderParams(**vo.FULL_PARAMS))
    sd = vo.synth_state_dict({k: v.shape for k, v in ae.state_dict().items()}, seed=7)
    ae.load_state_dict(sd, strict=True)
    ae.to(dev)
    z, x = vo.full_inputs()
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    out = ae.decode(z.to(dev)).float().cpu()
    assert out.shape == g["dec_ref_fp32"].shape and torch.isfinite(out).all()
    e_nat, e_ref = rel(out, g["dec_ref_fp32"]), rel(g["dec_ref_autocast"], g["dec_ref_fp32"])
    mom = ae.encode_moments(x.to(dev)).float().cpu()
    assert mom.shape == g["enc_moments_fp32"].shape and torch.isfinite(mom).all()
    m_nat, m_ref = rel(mom, g["enc_moments_fp32"]), rel(g["enc_moments_autocast"], g["enc_moments_fp32"])
    print(f"full-size VAE: decode native vs fp32 reference {e_nat:.3e} (reference autocast {e_ref:.3e}); encode moments {m_nat:.3e} "
          f"(reference autocast {m_ref:.3e})")
    assert e_nat <= 1.5 * e_ref and m_nat <= 1.5 * m_ref


def test_pipeline_img2img_through_vae_encoder(dev):
    """generate(init_image=..., strength=...) on the drop-in surface (reference flux_pipeline.py:459-523, 583-603): the init image is
    resized / centre-cropped, VAE-encoded natively, blended with the noise at t = timesteps[int((1 - strength) * num_steps)] and the
    loop runs only the remaining steps."""
    import io

    import numpy as np
    from PIL import Image

    from flux_pipeline import FluxPipeline
    from fluxmi import synth
    from modules.autoencoder import AutoEncoder, AutoEncoderParams

    cfg = tiny_config()
    cfg.text_enc_max_length = 32
    cfg.ae_device = str(dev)
    cfg.ae_params = AutoEncoderParams(resolution=32, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2, 2, 2], num_res_blocks=1, z_channels=16,
                                      scale_factor=0.3611, shift_factor=0.1159)
    torch.manual_seed(0)
    ae_sd = {k: v.clone() for k, v in AutoEncoder(cfg.ae_params).state_dict().items()}
    pipe = FluxPipeline.load_pipeline_from_config(cfg, state_dict=synth.make_state_dict(cfg.params, seed=0), ae_state_dict=ae_sd)
    assert pipe.ae is not None and pipe.ae.encoder_loaded
    pipe.compile()
    g = torch.Generator().manual_seed(1)
    prompt = {"txt": 0.1 * torch.randn(1, 32, 128, generator=g), "vec": torch.randn(1, 64, generator=g)}
    rng = np.random.default_rng(0)
    init = rng.integers(0, 256, size=(80, 120, 3), dtype=np.uint8)  # HWC, needs resize + crop to 96 x 64
    # schedule / blend bookkeeping
    x, ts = pipe.preprocess_latent(init_image=torch.from_numpy(init), height=96, width=64, num_steps=8, strength=0.5,
                                   generator=torch.Generator(device=dev).manual_seed(3), num_images=2)
    full = pipe.get_schedule(8, 12 * 8 // 4, shift=True)
    assert x.shape == (2, 16, 12, 8) and len(ts) == len(full) - 4 and ts[0] == full[4]
    # strength 1.0 -> the init image has no weight: identical to txt2img noise
    x1, ts1 = pipe.preprocess_latent(init_image=torch.from_numpy(init), height=96, width=64, num_steps=8, strength=1.0,
                                     generator=torch.Generator(device=dev).manual_seed(3), num_images=1)
    x0, ts0 = pipe.preprocess_latent(height=96, width=64, num_steps=8, generator=torch.Generator(device=dev).manual_seed(3), num_images=1)
    assert ts1 == ts0 and torch.equal(x1, x0)
    for src in (init, Image.fromarray(init), torch.from_numpy(init)):
        buf = pipe.generate(prompt, width=64, height=96, num_steps=8, seed=7, silent=True, init_image=src, strength=0.5)
        assert isinstance(buf, io.BytesIO)
        im = Image.open(buf)
        assert im.size == (64, 96) and im.mode == "RGB"


def test_pipeline_generate_jpeg_through_vae(dev):
    """generate() end to end on the drop-in surface: denoise loop -> unpack -> native VAE decode -> JPEG bytes
    (reference flux_pipeline.py:619-663, 423-448, 373-421)."""
    import io
    import os

    from PIL import Image
    from safetensors.torch import load_file

    from flux_pipeline import FluxPipeline
    from fluxmi import synth
    from modules.autoencoder import AutoEncoderParams

    cfg = tiny_config()
    cfg.text_enc_max_length = 32
    cfg.ae_device = str(dev)
    # the golden decoder has z_channels = 4; the flow model's 16 latent channels -> build a 16-channel decoder from it
    cfg.ae_params = AutoEncoderParams(resolution=32, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2, 2, 2], num_res_blocks=1, z_channels=16,
                                      scale_factor=0.3611, shift_factor=0.1159)
    from modules.autoencoder import AutoEncoder

    torch.manual_seed(0)
    ae_sd = {k: v.clone() for k, v in AutoEncoder(cfg.ae_params).state_dict().items()}
    sd = synth.make_state_dict(cfg.params, seed=0)
    pipe = FluxPipeline.load_pipeline_from_config(cfg, state_dict=sd, ae_state_dict=ae_sd)
    assert pipe.ae is not None
    pipe.compile()
    g = torch.Generator().manual_seed(1)
    prompt = {"txt": 0.1 * torch.randn(1, 32, 128, generator=g), "vec": torch.randn(1, 64, generator=g)}
    buf = pipe.generate(prompt, width=64, height=96, num_steps=4, seed=7, silent=True)
    assert isinstance(buf, io.BytesIO)
    im = Image.open(buf)
    assert im.size == (64, 96) and im.mode == "RGB"


def test_prequantized_checkpoint_round_trip(dev, tmp_path):
    """a calibrated model's state_dict() saved as safetensors (`float8_data / scale / input_scale ...`, reference
    float8_quantize.py:91-193, main.py:121-131) loads through `prequantized_flow=True` + `ckpt_path` without re-quantising or
    re-calibrating and reproduces the same latents bit for bit; the file is about half the bf16 checkpoint."""
    from safetensors.torch import save_file

    from flux_pipeline import FluxPipeline
    from fluxmi import synth

    cfg = tiny_config()
    cfg.text_enc_max_length = 32
    sd = synth.make_state_dict(cfg.params, seed=0)
    pipe = FluxPipeline.load_pipeline_from_config(cfg, state_dict=sd)
    pipe.compile()
    g = torch.Generator().manual_seed(1)
    prompt = {"txt": 0.1 * torch.randn(1, 32, 128, generator=g), "vec": torch.randn(1, 64, generator=g)}
    a = pipe.generate(prompt, width=64, height=96, num_steps=4, seed=3, silent=True)
    ck = {k: v.detach().cpu().contiguous() for k, v in pipe.model.state_dict().items()}
    assert any(k.endswith("float8_data") for k in ck) and any(k.endswith("input_scale") for k in ck)
    path = str(tmp_path / "flow-prequantized.safetensors")
    save_file(ck, path)
    bf16_bytes = sum(v.numel() * 2 for v in sd.values())
    assert os.path.getsize(path) < 0.62 * bf16_bytes

    cfg2 = tiny_config()
    cfg2.text_enc_max_length = 32
    cfg2.prequantized_flow = True
    cfg2.ckpt_path = path
    pipe2 = FluxPipeline.load_pipeline_from_config(cfg2)
    ok, _ = pipe2.model.calibration_state()
    assert ok  # input scales came from the file: no warm-up needed
    pipe2.compile()  # a no-op for prequantised checkpoints (reference flux_pipeline.py:197)
    for n, m in pipe.model.named_modules():
        if hasattr(m, "float8_data") and m.float8_data is not None:
            m2 = pipe2.model.get_submodule(n)
            assert torch.equal(m.float8_data.view(torch.uint8), m2.float8_data.view(torch.uint8)), n
            assert m.input_scale.item() == m2.input_scale.item() and m.scale.item() == m2.scale.item(), n
    b = pipe2.generate(prompt, width=64, height=96, num_steps=4, seed=3, silent=True)
    assert torch.equal(a, b)


def _reference_config_paths():
    import glob

    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs", "*.json")))


@pytest.mark.parametrize("path", _reference_config_paths(), ids=lambda p: os.path.basename(p))
def test_reference_config_jsons_load_and_generate(dev, path, tmp_path):
    """Drop-in check on the reference's OWN shipped JSONs (tests/golden/configs = /root/reference/configs): every file goes through
    FluxPipeline.load_pipeline_from_config_path with no dtype / flag override -- they all say flow_dtype float16, which the engine serves
    in bf16 with float16 at the model boundary -- and then denoises.  Only `params` is shrunk (hidden 256, 2+2 blocks: eleven 12 B-parameter
    loads are not a unit test; bench.py loads the full size) and the checkpoint comes from memory; compile_blocks / compile_extras configs
    run the reference's 768x768 warm-up inside the constructor, the prequantised one loads fp8 bytes + scales from a file."""
    import json
    import warnings

    from safetensors.torch import save_file

    import util
    from flux_pipeline import FluxPipeline
    from fluxmi import synth

    spec = json.load(open(path))
    spec["params"].update(hidden_size=256, num_heads=2, depth=2, depth_single_blocks=2, context_in_dim=128, vec_in_dim=64)
    spec["text_enc_max_length"] = 32
    tiny = util.ModelSpec(**spec)
    sd = synth.make_state_dict(tiny.params, seed=3)
    kw = {}
    if spec.get("prequantized_flow"):
        # what the reference's `--save-prequantized` produces (main.py:121-131): the calibrated state dict of the same model
        src = dict(spec, prequantized_flow=False, compile_blocks=False, compile_extras=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p0 = FluxPipeline.load_pipeline_from_config(util.ModelSpec(**src), state_dict={k: v.clone() for k, v in sd.items()})
            p0.compile()
        ck = str(tmp_path / "prequant.safetensors")
        save_file({k: v.detach().cpu().contiguous() for k, v in p0.model.state_dict().items()}, ck)
        spec["ckpt_path"] = ck
        del p0
    else:
        kw["state_dict"] = {k: v.clone() for k, v in sd.items()}
    cfg_file = tmp_path / os.path.basename(path)
    cfg_file.write_text(json.dumps(spec))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pipe = FluxPipeline.load_pipeline_from_config_path(str(cfg_file), **kw)
    assert pipe.config.flow_dtype == "float16" and pipe.dtype == torch.float16 and pipe.model.dtype == torch.float16
    if not (spec.get("compile_blocks") or spec.get("compile_extras") or spec.get("prequantized_flow")):
        pipe.compile()
    # flux-schnell's warm-up is 3 x 4 steps = 12 calls (reference flux_pipeline.py:205-209): the 13th call, which freezes the input
    # scales, is the first step of the first request
    assert pipe.model.calibration_state()[0] or tiny.version == "flux-schnell"
    g = torch.Generator().manual_seed(1)
    prompt = {"txt": 0.1 * torch.randn(1, 32, 128, generator=g), "vec": torch.randn(1, 64, generator=g)}
    steps = 4
    out = pipe.generate(prompt, width=64, height=96, num_steps=steps, seed=11, output_type="latent", silent=True)
    assert out.shape == (1, 16, 12, 8) and torch.isfinite(out).all() and pipe.model.calibration_state()[0]
    # Flux.forward hands back the configured dtype; the parameters the engine binds are bf16 / fp8
    d = {k: v.to(dev) for k, v in synth.make_inputs(tiny.params, 64, 96, 32, batch=1, seed=2).items()}
    pred = pipe.model(d["img"].half(), d["img_ids"].half(), d["txt"].half(), d["txt_ids"].half(), torch.full((1,), 0.5, device=dev).half(),
                      d["y"].half(), torch.full((1,), 3.5, device=dev).half() if tiny.params.guidance_embed else None)
    assert pred.dtype == torch.float16 and torch.isfinite(pred).all()


def test_large_batches_in_one_pass_and_in_equal_passes(dev):
    """The reference has no num_images limit.  The engine takes up to 32 samples per pass (round 2: 8): a batch of 10 in ONE pass must give every
    sample exactly what it gets alone (samples never interact; the embedder GEMV runs in row chunks of 8), and a batch above the limit
    runs as EQUAL consecutive passes (35 -> 12 + 12 + 11 padded to 12, one workspace / one graph) with the same bits.
    A calibrating model refuses the multi-pass form (its trial counters advance once per step, not once per pass)."""
    from fluxmi import synth

    cfg = tiny_config()
    model, _, _ = build(cfg, QUANTS["fp8"], dev)
    H, W, Lt = 64, 64, 32
    ts = fo.get_schedule(3, (H // 16) * (W // 16))
    inp = to_dev(synth.make_inputs(cfg.params, H, W, Lt, batch=35, seed=9, real_tokens=8), dev)
    sl = lambda n: tuple(inp[k][:n].contiguous() for k in ("img", "img_ids", "txt", "txt_ids", "y"))
    model.denoise(*sl(1), fo.get_schedule(13, (H // 16) * (W // 16)), guidance=3.5)  # calibration on sample 0
    assert model.calibration_state()[0]
    ten = model.denoise(*sl(10), ts, guidance=3.5)
    assert ten.shape[0] == 10 and torch.isfinite(ten).all()
    alone = [model.denoise(*(t[i:i + 1].contiguous() for t in sl(10)), ts, guidance=3.5) for i in (0, 7, 8, 9)]
    for i, a in zip((0, 7, 8, 9), alone):
        assert torch.equal(ten[i].view(torch.int16), a[0].view(torch.int16)), f"sample {i} of a 10-batch differs from the same sample alone"
    model.MAX_ENGINE_BATCH = 12  # instance override: 35 samples -> three passes of 12 (the last one padded by one copy)
    try:
        big = model.denoise(*sl(35), ts, guidance=3.5)
    finally:
        del model.MAX_ENGINE_BATCH
    assert big.shape[0] == 35 and torch.equal(big[:10].view(torch.int16), ten.view(torch.int16))
    for i in (11, 12, 23, 24, 34):
        one = model.denoise(*(t[i:i + 1].contiguous() for t in sl(35)), ts, guidance=3.5)
        assert torch.equal(big[i].view(torch.int16), one[0].view(torch.int16)), f"sample {i} of the 3-pass batch"
    fresh, _, _ = build(cfg, QUANTS["fp8"], dev)
    fresh.MAX_ENGINE_BATCH = 12
    with pytest.raises(ValueError, match="frozen"):
        fresh.denoise(*sl(35), ts, guidance=3.5)


def test_serving_sequence_at_real_width_is_reproducible_and_memory_flat(dev):
    """What the reference's api.py does to one pipeline (api.py:54-122 -> flux_pipeline.py:526-651): requests of different resolutions and batch
    sizes, back to back, through ONE engine.  Every change of shape re-sizes the engine workspace and re-captures the hipGraph, and the kernel
    selection changes with it (1024^2: persistent GEMM + one workgroup per attention task; 768^2: 192-row tiles + the balanced attention grid;
    512^2 B = 2: single partial rounds).  A request must give the same bits every time it comes back, whatever ran in between, and device
    memory in use (library allocations + torch's pool, hipMemGetInfo) must not grow from cycle to cycle.  Hidden 3072, 24 heads, 1 + 2 blocks;
    the full-depth version with more shapes is tools/soak.py (profiles/r05_soak.txt)."""
    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8
    from fluxmi import synth

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
    p = cfg.params
    p.depth, p.depth_single_blocks = 1, 2
    requests = [(1024, 1024, 1, 4), (768, 768, 1, 4), (512, 512, 2, 3), (1024, 768, 1, 3)]
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=0, device=dev))
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=True, quantize_flow_embedder_layers=False)
        inputs = [to_dev(synth.make_inputs(p, h, w, 512, batch=b, seed=20 + i), dev) for i, (h, w, b, _) in enumerate(requests)]

        def run(i, n):
            d = inputs[i]
            return model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], fo.get_schedule(n, d["img"].shape[1]), guidance=3.5)

        run(0, 13)  # calibration: 12 trials + the freezing call
        assert model.calibration_state()[0]
        first, used = {}, []
        for cycle in range(3):
            for i, (h, w, b, n) in enumerate(requests):
                out = run(i, n)
                assert torch.isfinite(out.float()).all(), f"cycle {cycle}, {h}x{w} B={b}: non-finite latents"
                bits = out.view(torch.int16)
                if cycle == 0:
                    first[i] = bits.clone()
                else:
                    assert torch.equal(bits, first[i]), f"cycle {cycle}, {h}x{w} B={b}: latents differ from the first time this request ran"
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            used.append(total - free)
        assert used[2] - used[1] <= (16 << 20) and used[1] - used[0] <= (64 << 20), f"device memory in use grows from cycle to cycle: {[u >> 20 for u in used]} MiB"
        del model


def test_one_engine_called_from_a_thread_pool(dev):
    """The reference's FastAPI handlers run in a thread pool (api.py:54: plain `def` endpoints) and share ONE pipeline without a lock; its eager torch ops
    tolerate that.  A C engine handle with a workspace and a captured hipGraph does not, so Flux.denoise serialises on a per-engine lock and the
    graph is captured in thread-local mode on a private stream (another thread's allocator calls must not poison the capture).  Three host
    threads issue different requests (two resolutions + a batch of two: every hand-over re-sizes the workspace and re-captures) against one
    frozen model, several times each; every result must equal the serial run's bits."""
    import threading

    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8
    from fluxmi import synth

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
    p = cfg.params
    p.depth, p.depth_single_blocks = 1, 2
    requests = [(1024, 1024, 1, 4), (768, 768, 1, 4), (512, 512, 2, 3)]
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=1, device=dev))
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=True, quantize_flow_embedder_layers=False)
        inputs = [to_dev(synth.make_inputs(p, h, w, 512, batch=b, seed=30 + i), dev) for i, (h, w, b, _) in enumerate(requests)]

        def run(i):
            d = inputs[i]
            return model.denoise(d["img"], d["img_ids"], d["txt"], d["txt_ids"], d["y"], fo.get_schedule(requests[i][3], d["img"].shape[1]), guidance=3.5)

        d0 = inputs[0]
        model.denoise(d0["img"], d0["img_ids"], d0["txt"], d0["txt_ids"], d0["y"], fo.get_schedule(13, d0["img"].shape[1]), guidance=3.5)  # calibration
        assert model.calibration_state()[0]
        serial = [run(i).view(torch.int16).clone() for i in range(len(requests))]
        torch.cuda.synchronize()
        errors, bar = [], threading.Barrier(len(requests))

        def work(i):
            try:
                with torch.inference_mode():
                    bar.wait()
                    for rep in range(6):
                        out = run(i)
                        torch.cuda.synchronize()
                        if not torch.equal(out.view(torch.int16), serial[i]):
                            errors.append(f"thread {i} repetition {rep}: latents differ from the serial run")
            except Exception as e:  # noqa: BLE001 -- reported by the main thread
                errors.append(f"thread {i}: {type(e).__name__}: {e}")

        th = [threading.Thread(target=work, args=(i,)) for i in range(len(requests))]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in th), "a worker thread hangs"
        assert not errors, errors
        del model


def test_maximum_batch_at_real_width(dev):
    """The engine's limit, B = 32 samples in ONE pass, at Flux-dev's width and 1024x1024 + 512 text tokens (1 + 1 blocks): the buffers cross
    2^31 bytes (linear1's output 4608 x 32 rows x 21504 bf16 = 6.3 GB, the linear2 input 2.3 G fp8 elements), every grouped launch carries 64
    (sample, stream) groups, attention runs 13 824 tasks.  Samples never interact (flux_model.py has no cross-sample op), so each of a handful of
    samples must come out of the batch exactly as it comes out alone -- any 32-bit offset, group-table or grid-size limit shows up as a mismatch."""
    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8
    from fluxmi import synth

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16", quantize_modulation=True, quantize_flow_embedder_layers=False)
    p = cfg.params
    p.depth, p.depth_single_blocks = 1, 1
    B = 32
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=2, device=dev))
        quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                      quantize_modulation=True, quantize_flow_embedder_layers=False)
        assert B == model.MAX_ENGINE_BATCH
        inp = to_dev(synth.make_inputs(p, 1024, 1024, 512, batch=B, seed=40), dev)
        keys = ("img", "img_ids", "txt", "txt_ids", "y")
        Li = inp["img"].shape[1]
        one = lambda i: tuple(inp[k][i:i + 1].contiguous() for k in keys)
        model.denoise(*one(0), fo.get_schedule(13, Li), guidance=3.5)  # calibration on sample 0
        assert model.calibration_state()[0]
        ts = fo.get_schedule(2, Li)
        whole = model.denoise(*(inp[k] for k in keys), ts, guidance=3.5)
        torch.cuda.synchronize()
        assert whole.shape[0] == B and torch.isfinite(whole.float()).all()
        for i in (0, 1, 14, 15, 16, 30, 31):
            alone = model.denoise(*one(i), ts, guidance=3.5)
            assert torch.equal(alone[0].view(torch.int16), whole[i].view(torch.int16)), f"sample {i} of the 32-batch differs from the same sample alone"
        del model


@pytest.mark.parametrize("flow", ["fp8_768", "bf16_schnell_256"])
def test_a_sample_does_not_depend_on_its_batch(flow, dev):
    """Round 6 (VERDICT r05 weak #5 / ADVICE): samples never interact in the reference (flux_model.py:672-716), so sample i of a batch of 2 or 4
    must come out with exactly the bits it has alone, at the two shapes where round 5 could not promise it:
      fp8_768          Flux-dev width, 768x768 + 512 text tokens (L = 2816): the balanced attention grid of the thin last round was planned
                       over the whole launch, so the key pieces of a (head, row block) followed B (<= 2.5e-3); now planned per sample
      bf16_schnell_256 the bf16 FLOW (no F8Linear; BASELINE configs[0]: schnell 256x256 + 256 text tokens, M = 512 rows per sample): the
                       dispatcher's tile / split-K choice followed the row count; now every bf16 tile config gives the same bits and the
                       split-K slices are decided on one sample's groups (csrc/api.cpp, fluxmi_gemm_set_batch)
    1 + 1 blocks at hidden 3072, frozen scales, a 2-step hipGraph loop."""
    import util
    from float8_quantize import quantize_flow_transformer_and_dispatch_float8
    from fluxmi import synth

    fp8 = flow == "fp8_768"
    cfg = util.load_config(util.ModelVersion.flux_dev if fp8 else util.ModelVersion.flux_schnell, flow_dtype="bfloat16", quantize_modulation=True,
                           quantize_flow_embedder_layers=False)
    p = cfg.params
    p.depth, p.depth_single_blocks = 1, 1
    side, lt = (768, 512) if fp8 else (256, 256)
    with torch.inference_mode():
        model = util.load_flow_model(cfg, synth.make_state_dict(p, seed=4, device=dev)).to(dev)
        if fp8:
            quantize_flow_transformer_and_dispatch_float8(model, dev, flow_dtype=torch.bfloat16, swap_linears_with_cublaslinear=False,
                                                          quantize_modulation=True, quantize_flow_embedder_layers=False)
        else:
            model.eval().requires_grad_(False)
            assert len(model.f8_modules()) == 0
        inp = to_dev(synth.make_inputs(p, side, side, lt, batch=4, seed=41), dev)
        keys = ("img", "img_ids", "txt", "txt_ids", "y")
        Li = inp["img"].shape[1]
        sl = lambda a, b: tuple(inp[k][a:b].contiguous() for k in keys)
        g = 3.5  # schnell has no guidance embedder: ignored there
        shift = fp8  # schnell: no time shift
        if fp8:
            model.denoise(*sl(0, 1), fo.get_schedule(13, Li), guidance=g)  # calibration on sample 0
            assert model.calibration_state()[0]
        ts = fo.get_schedule(2, Li, shift=shift)
        alone = [model.denoise(*sl(i, i + 1), ts, guidance=g)[0].clone() for i in range(4)]
        # hipGraph replay == eager launches at this shape too (768^2: the balanced attention grid hands partial softmax states between workgroups
        # through one XCD's L2 and relies on every dispatch -- graph-replayed or not -- starting with an L1 invalidate)
        eager = model.denoise(*sl(0, 1), ts, guidance=g, use_graph=False)[0]
        assert torch.equal(eager.view(torch.int16), alone[0].view(torch.int16)), f"{flow}: graph replay differs from eager launches"
        for B in (2, 4):
            whole = model.denoise(*sl(0, B), ts, guidance=g)
            torch.cuda.synchronize()
            assert whole.shape[0] == B and torch.isfinite(whole.float()).all()
            for i in range(B):
                same = torch.equal(alone[i].view(torch.int16), whole[i].view(torch.int16))
                assert same, f"{flow}: sample {i} of a batch of {B} differs from the same sample alone (rel-L2 {rel_l2(whole[i], alone[i]):.3e})"
        del model
