"""CPU, world_size 2 over gloo: the only multi-GPU exchanges of the path -- one flat broadcast of
the T5/CLIP embeddings + noise before the loop, batch sharding, gather of the latents, MAX all-reduce of calibration
amax values.  The same code runs over backend "nccl" (= RCCL/xGMI) on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "flux-fp8-api_amd"))
    from fluxmi import dist as fdist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = fdist.init_from_env("gloo")
    assert (r, w) == (rank, world) and fdist.is_dist()
    g = torch.Generator().manual_seed(0)
    txt = torch.randn(batch, 6, 16, generator=g).bfloat16()
    vec = torch.randn(batch, 8, generator=g).bfloat16()
    noise = torch.randn(batch, 12, 4, generator=g).bfloat16()
    ref = (txt.clone(), vec.clone(), noise.clone())
    if rank != 0:  # only the text-encoder rank holds the conditioning
        txt, vec, noise = torch.zeros_like(txt), torch.zeros_like(vec), torch.zeros_like(noise)
    txt, vec, noise = fdist.broadcast_request(txt, vec, noise, src=0)
    ok = all(torch.equal(a, b) for a, b in zip((txt, vec, noise), ref))
    lo, hi = fdist.shard_bounds(batch, rank, world)
    # "denoise" the local shard: a per-sample function, no cross-sample interaction
    local = noise[lo:hi] * 2 + txt[lo:hi].float().mean(dim=(1, 2), keepdim=True).bfloat16()
    full = fdist.gather_latents(local.contiguous(), batch, dst=0)
    if rank == 0:
        expect = ref[2] * 2 + ref[0].float().mean(dim=(1, 2), keepdim=True).bfloat16()
        ok = ok and torch.equal(full, expect)
    else:
        ok = ok and full is None
    am = torch.tensor([1.0 + rank, 5.0 - rank, 0.5])
    fdist.allreduce_amax(am)
    ok = ok and am.tolist() == [float(world), 5.0, 0.5]

    # calibration sync: every replica ends with the scales of the union of the batch (float8_quantize.py:227,237-246)
    class L:  # the attributes sync_calibration touches on an F8Linear
        def __init__(self, trials, max_value):
            self.input_amax_trials = trials
            self.input_scale = torch.ones(())
            self.input_scale_reciprocal = torch.ones(())
            self.input_max_value = max_value

    layers = [L(torch.tensor([0.5, 3.0 + rank, 1.0]), 57344.0), L(torch.tensor([1e-14 * (rank + 1), 0.0, 0.0]), 448.0)]
    ok = ok and fdist.sync_calibration(layers) == 2
    top = 3.0 + (world - 1)
    ok = ok and layers[0].input_amax_trials.tolist() == [0.5, top, 1.0]
    ok = ok and layers[0].input_scale.item() == (torch.tensor(57344.0) / torch.tensor(top)).item()
    ok = ok and layers[0].input_scale_reciprocal.item() == layers[0].input_scale.reciprocal().item()
    ok = ok and layers[1].input_scale.item() == 448.0  # clamped at the format maximum
    q.put((rank, bool(ok), (lo, hi)))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("batch", [2, 3, 8])
def test_two_rank_request_broadcast_shard_gather(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    (lo0, hi0), (lo1, hi1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == batch and abs((hi0 - lo0) - (hi1 - lo1)) <= 1


def test_shard_bounds_cover_and_balance():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "flux-fp8-api_amd"))
    from fluxmi.dist import shard_bounds

    for batch in (1, 2, 7, 8, 9, 64):
        for world in (1, 2, 4, 8):
            b = [shard_bounds(batch, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == batch
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(8, 3, 8) == (3, 4)  # configs[3]: batch 8, one image per GPU


def _pipeline_worker(rank, world, port, batch, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "flux-fp8-api_amd"))
    from flux_pipeline import FluxPipeline
    from fluxmi import dist as fdist

    class StubFlow:  # per-sample function of (latent tokens, conditioning, schedule): no cross-sample interaction, like Flux.forward
        calls = []

        def denoise(self, img, img_ids, txt, txt_ids, vec, timesteps, guidance=3.5, use_graph=True):
            StubFlow.calls.append(img.shape[0])
            assert img_ids.shape[0] == txt.shape[0] == txt_ids.shape[0] == vec.shape[0] == img.shape[0]
            return (img.float() * 2 + txt.float().mean(dim=(1, 2), keepdim=True) + vec.float().sum(-1)[:, None, None] + len(timesteps)).to(img.dtype)

    def make_pipe():
        pipe = FluxPipeline.__new__(FluxPipeline)
        pipe.name, pipe.debug, pipe.dtype, pipe.ae_dtype = "flux-dev", False, torch.bfloat16, torch.bfloat16
        pipe.device_flux = pipe.device_ae = pipe.device_clip = pipe.device_t5 = torch.device("cpu")
        pipe.model, pipe.ae, pipe.clip, pipe.t5, pipe.rng = StubFlow(), None, None, None, torch.Generator(device="cpu")
        return pipe

    g = torch.Generator().manual_seed(1)
    prompt = {"txt": torch.randn(batch, 6, 16, generator=g), "vec": torch.randn(batch, 8, generator=g)}
    kw = dict(width=64, height=96, num_steps=5, seed=11, num_images=batch, output_type="latent", silent=True)
    expect = make_pipe().generate(prompt, **kw)  # single process: the whole batch on one replica
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    fdist.init_from_env("gloo")
    StubFlow.calls.clear()
    if rank != 0:  # only rank 0 owns the real conditioning; the broadcast must overwrite whatever the others hold
        prompt = {k: torch.zeros_like(v) for k, v in prompt.items()}
    out = make_pipe().generate(prompt, **kw)
    lo, hi = fdist.shard_bounds(batch, rank, world)
    ok = StubFlow.calls == ([hi - lo] if hi > lo else [])  # a rank with an empty shard skips the denoise but still joins the gather
    ok = ok and ((out is not None and torch.equal(out, expect)) if rank == 0 else out is None)
    q.put((rank, bool(ok), (lo, hi)))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("batch", [1, 2, 5])
def test_two_rank_pipeline_generate_matches_single_process(batch):
    """FluxPipeline.generate under a 2-rank process group (gloo): embeddings + noise broadcast from rank 0, every rank denoises its own
    batch slice, rank 0 gathers -- and gets exactly what a single replica produces for the whole batch; other ranks return None.
    batch 1 < world 2: the rank with the empty shard must not raise or hang the collective (the API default is num_images = 1)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_calibration_guard_counts_a_list_prompt_as_its_batch(monkeypatch):
    """While the input scales are still moving, a request with fewer images than ranks is refused (a rank with an empty shard would not
    advance its trial counters) -- but the batch a LIST prompt with num_images == 1 builds is its length (flux_pipeline.py:267-278):
    two prompts on two ranks give every rank a shard and must pass the guard."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "flux-fp8-api_amd"))
    import flux_pipeline
    from flux_pipeline import FluxPipeline

    class Reached(Exception):
        pass

    class CalibratingFlow:
        def calibration_state(self):
            return (False, 3)

    pipe = FluxPipeline.__new__(FluxPipeline)
    pipe.name, pipe.debug, pipe.dtype, pipe.ae_dtype = "flux-dev", False, torch.bfloat16, torch.bfloat16
    pipe.device_flux = pipe.device_ae = pipe.device_clip = pipe.device_t5 = torch.device("cpu")
    pipe.model, pipe.ae, pipe.clip, pipe.t5, pipe.rng = CalibratingFlow(), None, None, None, torch.Generator(device="cpu")

    def reached(*a, **k):
        raise Reached()

    pipe.prepare = reached
    monkeypatch.setattr(flux_pipeline.fdist, "world_size", lambda: 2)
    monkeypatch.setattr(flux_pipeline.fdist, "rank", lambda: 0)
    kw = dict(width=64, height=64, num_steps=2, seed=1, output_type="latent", silent=True)
    with pytest.raises(RuntimeError, match="fewer images"):
        pipe.generate("one prompt", num_images=1, **kw)
    with pytest.raises(RuntimeError, match="fewer images"):
        pipe.generate(["only one"], num_images=1, **kw)
    with pytest.raises(Reached):
        pipe.generate(["first", "second"], num_images=1, **kw)
    with pytest.raises(Reached):
        pipe.generate("one prompt", num_images=2, **kw)


# ---- bench.py: the launch paths the driver uses for N > 1, on CPU (gloo, stub engine) ---------------------------------------------------
def _run_bench(extra, env=None, launcher=False, timeout=240):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    args = ["--backend", "gloo", "--dry-run", "--steps", "6", "--warmup", "2"] + extra
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench] + args
    else:
        cmd = [sys.executable, bench] + args
    e = dict(os.environ, OMP_NUM_THREADS="2")
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


@pytest.mark.parametrize("launcher", [False, True], ids=["python bench.py --gpus 2", "torch.distributed.run ... bench.py --gpus 2"])
def test_bench_multi_rank_launch_paths(launcher):
    """`python bench.py --gpus 2` must launch its own ranks (what the round-2 driver ran: it exited with `launch with torch.distributed.run`),
    and the documented torchrun form must keep working.  Stub engine over gloo: rendezvous on 127.0.0.1, ONE broadcast of the request, batch
    sharding, 13 calibrating steps with the in-step amax exchange, barrier + max-over-ranks timing, exactly ONE JSON line from rank 0 with
    the world size and backend the process group reports."""
    r, out = _run_bench(["--gpus", "2"], launcher=launcher)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1, r.stdout
    d = out[0]
    assert d["n_gpus"] == 2 and d["config"]["nranks"] == 2 and d["config"]["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["dry_run"] and d["amax_exchanges"] == 13 and d["config"]["calibration"].startswith("in-step")
    assert d["steps"] == 6 and d["warmup"] == 2 and d["config"]["finite_output"] and d["value"] > 0
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) / d["value"] < 1e-2  # whole-job aggregate


def test_bench_single_rank_dry_run_and_failed_exchange():
    r, out = _run_bench(["--gpus", "1"])
    assert r.returncode == 0 and len(out) == 1 and out[0]["n_gpus"] == 1 and out[0]["config"]["backend"] is None, r.stderr[-2000:]
    # the headline is the MEDIAN of `requests` timed repeats of `steps` steps each; every repeat and the range are in the line
    d = out[0]
    each = d["ms_per_step_each"]
    assert d["requests"] == len(each) == 3 and d["ms_per_step"] == sorted(each)[1]
    assert d["value_range"][0] <= d["value"] <= d["value_range"][1]
    # value = steps / time of the median request (ms_per_step is rounded to a microsecond: the stub engine's steps take ~20 of them)
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] <= 0.0006 / d["ms_per_step"] + 1e-3
    # a rank whose calibration exchange breaks must take the whole job down with a non-zero status (no silent per-rank fallback)
    r, out = _run_bench(["--gpus", "2"], env={"FLUXMI_BENCH_FAIL_RANK": "1"}, timeout=400)
    assert r.returncode != 0 and not out, (r.returncode, r.stdout[-500:])
    assert "injected exchange failure" in r.stderr


def test_bench_preflight_and_batched_config():
    """`bench.py --gpus N --preflight`: the first contact of the N-rank plumbing without a model -- process group, rank-id
    all-reduce, ONE broadcast of a request-sized payload (T5 states + CLIP vector + packed noise of every image), the latent gather -- one JSON
    line from rank 0, and a DISTINCT exit status per failing stage (10 rendezvous, 11 all-reduce, 12 broadcast, 13 gather).  And `--config 4`
    (BASELINE configs[3], batch 8): 8 / N images per rank through one engine each, value = image-steps per second of the whole job."""
    r, out = _run_bench(["--gpus", "2", "--preflight"])
    assert r.returncode == 0 and len(out) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = out[0]
    assert d["preflight"] == "ok" and d["nranks"] == 2 and d["backend"] == "gloo" and d["batch"] == 2 and d["total_s"] < 30
    assert d["broadcast_bytes"] == 2 * (512 * 4096 + 768 + 4096 * 64) * 2
    for stage, code in (("allreduce", 11), ("broadcast:1", 12), ("gather", 13), ("rendezvous", 10)):
        r, out = _run_bench(["--gpus", "2", "--preflight"], env={"FLUXMI_PREFLIGHT_FAIL": stage})
        assert r.returncode != 0 and not out and f'"stage": "{stage.split(":")[0]}"' in r.stderr, (stage, r.returncode, r.stderr[-800:])
        assert str(code) in r.stderr or r.returncode in (code, 1), (stage, r.returncode)  # torch.distributed.run reports the failing rank's exit code
    r, out = _run_bench(["--gpus", "1", "--preflight", "--single-rank-group"])
    assert r.returncode == 0 and out[0]["preflight"] == "ok" and out[0]["nranks"] == 1
    r, out = _run_bench(["--gpus", "2", "--config", "4"])
    assert r.returncode == 0 and len(out) == 1, r.stderr[-1500:]
    d = out[0]
    assert d["config"]["images_per_gpu"] == 4 and d["config"]["global_batch"] == 8 and d["n_gpus"] == 2
    assert abs(d["value"] - 8 * d["loop_its_per_gpu"]) / d["value"] < 1e-3 and d["image_steps_per_s"] == d["value"]


def test_bench_step_trace_and_step_pmc_summaries(tmp_path, monkeypatch):
    """bench.py's in-step roofline on synthetic rocprofv3 CSVs: the steady steps are the dispatches between the first and the last
    euler_kernel of the LAST request (the first step carries the request's set-up and is dropped), kernels are binned into GEMM / attention / LayerNorm
    / other, and the families + gaps sum to the wall time per step; the counter summary applies the gfx950 fetch correction (2 x FETCH_SIZE + WRITE_SIZE,
    KiB) and the matrix-pipe formula per family."""
    import csv
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    # ---- kernel trace: calibration, then a 2-step warm-up request, then a 4-step request (3 steady steps) ---------------------------------
    rows, t = [], [1000]

    def k(name, dur):
        rows.append(dict(Kernel_Name=name, Start_Timestamp=t[0], End_Timestamp=t[0] + dur * 1000))
        t[0] += dur * 1000 + 10000  # durations in us, 10 us between kernels

    def step():
        k("void gemm_ps_kernel<true, 1, 3, false>(FluxmiGemmParams)", 3000)
        k("void (anonymous namespace)::attention2_kernel<1, true, false, true>(AttnArgs)", 2000)
        k("ln_modulate_stream_kernel<6, true, 1, true>(LnModArgs, int, int)", 500)
        k("gemm_w1_kernel<true, 1, 2, 4>(FluxmiGemmParams)", 1500)
        k("select_step_kernel(int)", 50)
        k("euler_kernel(unsigned short*)", 40)

    for _ in range(3):
        k("amax_kernel(float*)", 30); k("calib_update_kernel(float*)", 20); step()
    for n_steps in (2, 4):
        k("timestep_rows_kernel(unsigned short*)", 25); k("build_qlut_kernel<1>(float const*)", 25)
        for _ in range(n_steps):
            step()
    d = tmp_path / "trace"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader(); w.writerows(rows)
    s = bench.summarize_step_trace(str(d), header="hdr")
    assert s["steps"] == 3 and s["launches"] == 6.0
    assert (s["gemm_ms"], s["attention_ms"], s["ln_ms"], s["other_ms"]) == (4.5, 2.0, 0.5, 0.09)
    assert abs(s["gemm_ms"] + s["attention_ms"] + s["ln_ms"] + s["other_ms"] + s["gaps_ms"] - s["wall_ms"]) < 2e-3 and abs(s["gaps_ms"] - 0.06) < 2e-3
    assert s["text"].startswith("hdr") and "gemm_ps_kernel" in s["text"] and "3 graph-replayed denoise steps of the last request" in s["text"]
    # one-step requests (config 1): every step carries its request's set-up, which then counts
    rows.clear(); t[0] = 1000
    for _ in range(6):
        k("timestep_rows_kernel(unsigned short*)", 25); step()
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader(); w.writerows(rows)
    s1 = bench.summarize_step_trace(str(d))
    assert s1["steps"] == 4 and s1["launches"] == 7.0 and "one-step requests" in s1["text"]

    # ---- counters: three passes, 2 steady steps ---------------------------------------------------------------------------------------------
    monkeypatch.setattr(bench, "step_pmc_file", lambda cfg: str(tmp_path / f"step_pmc_{cfg}.json"))
    names = ["timestep_rows_kernel(x)", "gemm_ps_kernel<1>(P)", "attention2_kernel<1>(A)", "euler_kernel(x)"] + ["gemm_ps_kernel<1>(P)", "attention2_kernel<1>(A)", "euler_kernel(x)"] * 2
    per = {"FETCH_SIZE": {"gemm": 1000.0, "attention": 100.0}, "WRITE_SIZE": {"gemm": 200.0, "attention": 50.0},
           "GRBM_GUI_ACTIVE": {"gemm": 8000.0, "attention": 8000.0}, "SQ_VALU_MFMA_BUSY_CYCLES": {"gemm": 512000.0, "attention": 256000.0}}
    for pi, counters in enumerate(bench.STEP_PMC_PASSES):
        pd = tmp_path / "pmc" / f"p{pi}"
        pd.mkdir(parents=True)
        with open(pd / "pmc_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            for di, n in enumerate(names):
                fam = "gemm" if "gemm" in n else "attention" if "attention" in n else None
                for c in counters:
                    w.writerow(dict(Dispatch_Id=di + 1, Kernel_Name=n, Counter_Name=c, Counter_Value=per[c].get(fam, 0.0) if fam else 0.0))
    p = bench.summarize_step_pmc(str(tmp_path / "pmc"), 2, ms_per_step=1.0)
    assert p["steps"] == 2
    g, a = p["per_step"]["gemm"], p["per_step"]["attention"]
    assert g["hbm_bytes"] == (2 * 1000 + 200) * 1024 and a["hbm_bytes"] == (2 * 100 + 50) * 1024
    assert g["mfma_busy_frac"] == 0.5 and a["mfma_busy_frac"] == 0.25 and p["mfma_busy_frac_step"] == 0.375
    assert abs(p["hbm_gbs_at_timed_ms_per_step"] - (g["hbm_bytes"] + a["hbm_bytes"]) / 1e-3 / 1e9) < 0.1
