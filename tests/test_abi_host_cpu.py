"""CPU (no GPU): the C ABI library loads and exports exactly what include/fluxmi.h declares, argument validation
returns errors instead of crashing, and the host-side mirror of the reference API behaves like the reference."""
import ctypes as C
import glob
import json
import math
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "fluxmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fluxmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from fluxmi import _lib

    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), f"libfluxmi.so does not export {s}"
    assert set(_lib.EXPORTS) == set(syms), f"ctypes table out of sync with the header: {set(_lib.EXPORTS) ^ set(syms)}"
    assert _lib.lib.fluxmi_abi_version() == 5


@pytest.mark.parametrize("B,L,H", [(1, 4608, 24), (1, 2816, 24), (2, 4608, 24), (8, 4608, 24), (1, 1100, 8), (3, 4608, 24), (1, 8192, 24), (1, 512, 24),
                                   (1, 4608, 3), (1, 1100, 16), (1, 6000, 24), (1, 1536, 24), (1, 2048, 24), (1, 3072, 24)])
def test_attention_balanced_grid_plan(B, L, H):
    """fluxmi_attention_plan (host arithmetic, no GPU): the pieces of the balanced attention grid tile every binned task's key range
    exactly once, piece bookkeeping (index, count, scratch slot) is consistent, and replaying the launch order on an XCD's 32 CUs -- each
    finished CU takes the next piece -- gives every CU one bin's worth of key tiles.  A thin last round (<= 8 of 32 CUs) is folded into the
    full round in front of it (768^2: 33 tasks over 32 bins), a single partial round is spread over all CUs, a fuller last round is binned on
    its own (1024^2: 22 tasks over 32 bins; only with attn_split = 2).                                      flux_model.py:41-45 is unchanged by any of it"""
    from fluxmi import ops

    plan = ops.attention_plan(B, L, H)
    # round 6: planned PER SAMPLE (a batch is launched sample by sample when the plan is on), so a sample's pieces never depend on its batch
    assert plan == ops.attention_plan(1, L, H)
    tasks, nt = (L + 255) // 256 * H, (L + 63) // 64
    n, last = tasks // 8, (tasks // 8) % 32
    if tasks % 8 or nt < 16 or last == 0 or last > 26:
        assert plan is None
        return
    expect = {(1, 4608, 24): (32, 22), (1, 2816, 24): (0, 33), (2, 4608, 24): (32, 22), (8, 4608, 24): (32, 22), (1, 1100, 8): (0, 5), (1, 1536, 24): (0, 18),
              (1, 3072, 24): (0, 36)}
    if (B, L, H) in expect:
        assert plan is not None and (plan["full_per_x"], plan["n_per_x"] - plan["full_per_x"]) == expect[(B, L, H)]
    if plan is None:
        return
    assert plan["thin"] == (n >= 32 and last <= 8)
    rem = plan["n_per_x"] - plan["full_per_x"]
    assert plan["n_per_x"] == n and plan["full_per_x"] % 32 == 0 and 1 <= rem <= 64
    ps = plan["pieces"]
    assert 1 <= len(ps) <= 64 and any(p["np"] > 1 for p in ps)
    by_task = {}
    for p in ps:
        assert p["len"] >= 1 and p["tb"] + p["len"] <= nt and p["tloc"] < rem
        by_task.setdefault(p["tloc"], []).append(p)
    assert sorted(by_task) == list(range(rem))
    slots = set()
    for t, pl in by_task.items():
        pl.sort(key=lambda p: p["tb"])
        assert [p["pidx"] for p in pl] == list(range(len(pl))) and all(p["np"] == len(pl) and p["base"] == pl[0]["base"] for p in pl) and len(pl) <= 8
        pos = 0
        for p in pl:  # contiguous, disjoint, complete
            assert p["tb"] == pos
            pos += p["len"]
            slot = p["base"] + p["pidx"]
            assert slot < 64 and slot not in slots
            slots.add(slot)
        assert pos == nt
    # list scheduling of the launch order on an XCD's 32 CUs
    free = [0] * 32
    for p in ps:
        i = min(range(32), key=lambda c: free[c])
        free[i] += p["len"]
    # one bin = 1/32 of the binned tiles, but never less than a quarter task (a task is cut into at most four bins) or 8 tiles; + the edge snap
    assert max(free) <= max(math.ceil(rem * nt / 32), math.ceil(nt / 4), 8) + 4, (max(free), rem * nt / 32)
    # the grid it replaces spends one whole task per CU on the last round (two for a thin round folded into the round in front of it)
    assert max(free) < (2 * nt if rem > 32 else nt)


def test_errors_are_returned_not_thrown():
    from fluxmi import _lib

    lib = _lib.lib
    assert lib.fluxmi_amax(None, None, 1, 7, 7, None) == 1 and b"multiples of 8" in lib.fluxmi_last_error()
    g = (_lib.GemmGroup * 1)()
    assert lib.fluxmi_gemm_grouped(g, 0, 128, 128, 1, 1, 0, -1, None) == 1
    assert lib.fluxmi_gemm_grouped(g, 1, 0, 128, 1, 1, 0, -1, None) == 1
    g[0].M = 5
    assert lib.fluxmi_gemm_grouped(g, 1, 128, 128, 1, 1, 0, -1, None) == 1 and b"NULL" in lib.fluxmi_last_error()
    assert lib.fluxmi_calib_update(None, None, None, None, 13, 12, 1.0, None) == 1
    d = _lib.ModelDesc()
    d.hidden, d.heads, d.depth, d.depth_single, d.guidance_embed = 256, 4, 1, 1, 1  # head_dim 64: unsupported
    d.axes_dim = (C.c_int * 3)(16, 56, 56)
    d.num_trials = 12
    n = lib.fluxmi_engine_num_linears(C.byref(d))
    assert n == 6 + 2 + 10 + 3 + 2
    lin = (_lib.Linear * n)()
    nrm = (C.c_void_p * 6)()
    h = C.c_void_p()
    assert lib.fluxmi_engine_create(C.byref(d), lin, n, nrm, 6, C.byref(h)) == 1 and b"head_dim" in lib.fluxmi_last_error()
    with pytest.raises(RuntimeError, match="fluxmi"):
        _lib.call("fluxmi_quantize_act", None, None, None, 1, 7, 7, 7, 1, None)


def test_tuning_struct_round_trip_and_validation(monkeypatch):
    """fluxmi_tuning_t (ABI 3): every kernel-selection knob lives in ONE struct that the library resolves once from the FLUXMI_*
    environment (csrc/tuning.cpp is the only getenv site) and that fluxmi_set_tuning replaces after validating every field."""
    from fluxmi import _lib

    src = ""
    for f in glob.glob(os.path.join(ROOT, "flux-fp8-api_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")) and not f.endswith("tuning.cpp"):
            src += open(f).read()
    assert "getenv" not in src, "kernel-selection knobs are read in csrc/tuning.cpp only"
    # the header's field list == the ctypes struct's
    hdr = open(os.path.join(ROOT, "include", "fluxmi.h")).read()
    body = hdr[hdr.index("typedef struct fluxmi_tuning {"):hdr.index("} fluxmi_tuning_t;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|float)\s+([a-z0-9_]+)\s*;", body)
    assert fields == [f[0] for f in _lib.Tuning._fields_]
    t = _lib.get_tuning()
    assert t.struct_size == C.sizeof(_lib.Tuning) and t.gemm_cfg == -1 and t.gemm_persist == 1 and t.ln_variant == 2
    assert abs(t.attn_defer_log2 - 8.0) < 1e-6 and t.attn_f16k == 1 and t.fuse_kv == 2 and t.qlut == 1
    with _lib.tuning(gemm_cfg=13, attn_defer_log2=6.5):
        u = _lib.get_tuning()
        assert u.gemm_cfg == 13 and abs(u.attn_defer_log2 - 6.5) < 1e-6
    assert _lib.get_tuning().gemm_cfg == -1
    for bad in (dict(attn_defer_log2=float("nan")), dict(attn_defer_log2=-1.0), dict(attn_defer_log2=100.0), dict(fuse_kv=7), dict(ln_variant=0)):
        with pytest.raises(RuntimeError, match="tuning"):
            _lib.set_tuning(**bad)
    assert _lib.get_tuning().fuse_kv == 2  # a refused struct changes nothing
    bad = _lib.get_tuning()
    bad.struct_size = 4
    assert _lib.lib.fluxmi_set_tuning(C.byref(bad)) == 1 and b"struct_size" in _lib.lib.fluxmi_last_error()
    with pytest.raises(KeyError):
        _lib.set_tuning(no_such_knob=1)


def test_struct_layouts_match_the_header():
    from fluxmi import _lib

    assert C.sizeof(_lib.GemmGroup) == 10 * 8 + 4 * 8 + 4 * 4 + 4 * 8 + 8 + 6 * 4 + 8 + 8 + 2 * 4  # ... q_lut, W_pairs, a_pairs, c8_pairs
    assert C.sizeof(_lib.Linear) == 6 * 8 + 4 * 4
    assert C.sizeof(_lib.ModelDesc) == 14 * 4


def test_ops_refuse_cpu_tensors():
    from fluxmi import ops

    with pytest.raises(RuntimeError, match="GPU"):
        ops.quantize_act(torch.zeros(2, 8, dtype=torch.bfloat16), torch.ones(()))


# ---- config surface (reference util.py:38-79, 122-222) ----------------------------------------------------------
def test_modelspec_defaults_and_load_config():
    import util

    cfg = util.load_config(util.ModelVersion.flux_dev)
    assert cfg.quantize_modulation is True and cfg.quantize_flow_embedder_layers is False
    assert cfg.flow_dtype == "float16" and cfg.text_enc_max_length == 512 and cfg.params.guidance_embed
    s = util.load_config(util.ModelVersion.flux_schnell)
    assert s.text_enc_max_length == 256 and not s.params.guidance_embed and s.repo_flow == "flux1-schnell.sft"
    assert (cfg.params.depth, cfg.params.depth_single_blocks, cfg.params.hidden_size, cfg.params.num_heads) == (19, 38, 3072, 24)
    with pytest.raises(ValueError):
        util.into_dtype("float64")
    with pytest.raises(ValueError):
        util.load_config_from_path("/nonexistent.json")
    assert util.into_device(1) == torch.device("cuda:1") and util.into_device(None) == torch.device("cuda:0")


def test_config_jsons_load(tmp_path):
    import util

    files = sorted(glob.glob(os.path.join(ROOT, "flux-fp8-api_amd", "configs", "*.json")))
    files += sorted(glob.glob("/root/reference/configs/*.json"))  # present in the build container only
    assert files
    for f in files:
        cfg = util.load_config_from_path(f)
        assert cfg.params.hidden_size == 3072
    # unknown keys (the reference's own JSONs carry offload_ae / offload_text_enc) are ignored
    d = json.load(open(files[0]))
    d["offload_ae"] = True
    p = tmp_path / "c.json"
    p.write_text(json.dumps(d))
    assert util.load_config_from_path(str(p)).offload_vae is False


def test_flux_module_tree_matches_bfl_state_dict():
    import util
    from fluxmi import synth

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16")
    p = cfg.params
    p.hidden_size, p.num_heads, p.depth, p.depth_single_blocks, p.context_in_dim, p.vec_in_dim = 256, 2, 2, 3, 128, 64
    sd = synth.make_state_dict(p, seed=0)
    m = util.load_flow_model(cfg, sd)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert len(m.linear_modules()) == 6 + 2 + 2 * 10 + 3 * 3 + 2
    # walked-by-name attributes the reference relies on (float8_quantize.py:447-455, lora_loading.py:466-473)
    for name in ("double_blocks.1.img_attn.qkv", "double_blocks.0.txt_mlp.2", "single_blocks.2.linear2", "single_blocks.0.modulation.lin",
                 "final_layer.adaLN_modulation.1", "time_in.out_layer", "guidance_in.in_layer", "double_blocks.0.img_attn.norm.key_norm"):
        m.get_submodule(name)
    with pytest.raises(ValueError):
        bad = util.load_config(util.ModelVersion.flux_dev)
        bad.params.axes_dim = [16, 56, 40]
        util.load_flow_model(bad)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3), None, torch.zeros(1, 2, 3), None, None, None)


def test_prequantized_state_dict_format():
    """F8Linear._load_from_state_dict keeps the reference's checkpoint format (float8_quantize.py:91-193)."""
    from float8_quantize import F8Linear

    lin = F8Linear(16, 8, bias=True, dtype=torch.bfloat16)
    sd = {"weight": torch.zeros(1, dtype=torch.bfloat16), "bias": torch.ones(8, dtype=torch.bfloat16),
          "float8_data": torch.randn(8, 16).to(torch.float8_e4m3fn), "scale": torch.tensor(3.0), "scale_reciprocal": torch.tensor(1 / 3.0),
          "input_scale": torch.tensor(5.0), "input_scale_reciprocal": torch.tensor(0.2)}
    lin.load_state_dict(sd, assign=True)
    assert lin.weight_initialized and lin.input_scale_initialized and lin.trial_index == lin.num_scale_trials
    assert lin.scale.item() == 3.0 and lin.input_scale.item() == 5.0 and lin.weight.shape == (1,)
    with torch.device("meta"):  # util.load_flow_model builds the model on meta and assigns the checkpoint tensors
        lm = F8Linear(16, 8, bias=True, dtype=torch.bfloat16)
    lm.load_state_dict(sd, assign=True)
    lm.to("cpu")
    assert not lm.input_amax_trials.is_meta and all(not t.is_meta for t in list(lm.parameters()) + list(lm.buffers()))
    lin2 = F8Linear(16, 8, bias=True, dtype=torch.bfloat16)
    del sd["input_scale"], sd["input_scale_reciprocal"]
    lin2.load_state_dict(sd, assign=True)
    assert lin2.weight_initialized and not lin2.input_scale_initialized and lin2.trial_index == 0
    lin3 = F8Linear(16, 8, bias=True, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        lin3.load_state_dict({"weight": torch.zeros(3, 3, dtype=torch.bfloat16)}, assign=True)
    assert set(k for k, _ in lin.named_buffers()) >= {"float8_data", "scale", "scale_reciprocal", "input_scale", "input_scale_reciprocal"}


def test_pipeline_schedule_pack_unpack_match_oracle():
    import flux_oracle as fo
    from flux_pipeline import FluxPipeline

    pipe = FluxPipeline.__new__(FluxPipeline)
    assert pipe.get_schedule(28, 4096) == fo.get_schedule(28, 4096)
    assert pipe.get_schedule(4, 256, shift=False) == fo.get_schedule(4, 256, shift=False)
    x = torch.randn(2, 16, 8, 12)
    assert torch.equal(FluxPipeline.pack(x), fo.pack_latent(x))
    assert torch.equal(pipe.unpack(FluxPipeline.pack(x), 64, 96), x)
    ids = FluxPipeline.make_img_ids(2, 4, 6, "cpu", torch.bfloat16)
    assert torch.equal(ids, fo.make_ids(2, 4, 6, 3, torch.bfloat16)[0])


def test_kohya_lora_keys_convert():
    import lora_loading as ll

    sd = {"lora_unet_double_blocks_0_img_attn_qkv.lora_down.weight": torch.zeros(4, 8),
          "lora_unet_double_blocks_0_img_attn_qkv.lora_up.weight": torch.zeros(24, 4),
          "lora_unet_double_blocks_0_img_attn_qkv.alpha": torch.tensor(4.0),
          "lora_unet_single_blocks_11_linear2.lora_down.weight": torch.zeros(4, 8),
          "lora_unet_double_blocks_3_txt_mlp_2.lora_up.weight": torch.zeros(8, 4)}
    out = ll._kohya_to_bfl(sd)
    assert set(out) == {"double_blocks.0.img_attn.qkv.lora_A.weight", "double_blocks.0.img_attn.qkv.lora_B.weight",
                        "double_blocks.0.img_attn.qkv.alpha", "single_blocks.11.linear2.lora_A.weight", "double_blocks.3.txt_mlp.2.lora_B.weight"}
    assert ll._keys_without_ab(out) == ["double_blocks.0.img_attn.qkv", "double_blocks.3.txt_mlp.2", "single_blocks.11.linear2"]


def test_resize_center_crop_arithmetic():
    """FluxPipeline.resize_center_crop restates torchvision's TF.resize(size=int) + TF.center_crop (reference flux_pipeline.py:450-457):
    shorter edge -> min(width, height), long edge int(size * long / short), centred crop, zero pad where the resized image is smaller."""
    import torch

    from flux_pipeline import FluxPipeline

    img = torch.arange(3 * 80 * 120, dtype=torch.float32).reshape(1, 3, 80, 120) / (3 * 80 * 120)
    out = FluxPipeline.resize_center_crop(img, 96, 64)  # short edge 80 -> 64, long 120 -> 96; crop H 64 -> pad to 96
    assert out.shape == (1, 3, 96, 64)
    assert torch.all(out[..., :16, :] == 0) and torch.all(out[..., 80:, :] == 0) and out[..., 16:80, :].abs().sum() > 0
    same = FluxPipeline.resize_center_crop(img, 80, 120)  # already the right size: untouched
    assert torch.equal(same, img)
    crop = FluxPipeline.resize_center_crop(img, 80, 100)  # short edge already 80: crop only, origin round((120 - 100) / 2) = 10
    assert torch.equal(crop, img[..., :, 10:110])
    up = FluxPipeline.resize_center_crop(img.to(torch.bfloat16), 160, 160)  # bf16 in -> fp32 bilinear -> bf16 out, 160 x 240 -> crop
    assert up.shape == (1, 3, 160, 160) and up.dtype == torch.bfloat16


def test_http_api_contract():
    """api.py keeps the reference's endpoints, request schema, defaults and status codes (api.py:26-122), exercised with a stub model."""
    import io

    from fastapi.testclient import TestClient

    import api

    calls = []

    class Stub:
        def generate(self, **kw):
            calls.append(("generate", kw))
            return io.BytesIO(b"\xff\xd8jpeg-bytes\xff\xd9")

        def load_lora(self, path, scale, name):
            calls.append(("load", path, scale, name))
            if path == "missing.safetensors":
                raise FileNotFoundError("no such LoRA")

        def unload_lora(self, ident):
            calls.append(("unload", ident))

    api.app.state.model = Stub()
    c = TestClient(api.app)
    r = c.post("/generate", json={"prompt": "a (red:1.3) fox"})
    assert r.status_code == 200 and r.headers["content-type"] == "image/jpeg" and r.content.startswith(b"\xff\xd8")
    kw = calls[-1][1]
    assert kw["prompt"] == "a (red:1.3) fox" and (kw["width"], kw["height"], kw["num_steps"], kw["guidance"]) == (720, 1024, 24, 3.5)
    assert kw["strength"] == 1.0 and kw["init_image"] is None and 0 < kw["seed"] < api.MAX_RAND
    assert c.post("/generate", json={"prompt": "x", "seed": 0}).status_code == 422  # seed must be > 0
    assert c.post("/generate", json={}).status_code == 422
    r = c.post("/generate", json={"prompt": "x", "width": 512, "height": 512, "num_steps": 4, "seed": 7, "strength": 0.6, "init_image": "in.png"})
    assert r.status_code == 200 and calls[-1][1]["init_image"] == "in.png" and calls[-1][1]["seed"] == 7
    r = c.post("/lora", json={"path": "a.safetensors", "scale": 0.8, "name": "style"})
    assert r.status_code == 200 and r.json() == {"status": "success"} and calls[-1] == ("load", "a.safetensors", 0.8, "style")
    assert c.post("/lora", json={"action": "unload", "name": "style", "path": "a.safetensors"}).status_code == 200 and calls[-1] == ("unload", "style")
    assert c.post("/lora", json={"action": "unload", "path": "a.safetensors"}).status_code == 200 and calls[-1] == ("unload", "a.safetensors")
    r = c.post("/lora", json={"path": "missing.safetensors"})
    assert r.status_code == 500 and r.json() == {"status": "error", "message": "no such LoRA"}
    assert c.post("/lora", json={"action": "reload"}).status_code == 422


def test_cli_flags_match_reference():
    """main.py: the reference's flags, short names, defaults and inverted offload switches (main.py:7-148)."""
    import main

    a = main.parse_args([])
    assert (a.port, a.host, a.model_version, a.flux_device, a.num_to_quant, a.quant_text_enc) == (8088, "0.0.0.0", "flux-dev", "cuda:0", 20, "qfloat8")
    assert a.offload_ae and a.offload_text_enc and not a.offload_flow and a.quantize_modulation and not a.quantize_flow_embedder_layers
    a = main.parse_args(["-c", "cfg.json", "-p", "9000", "-OA", "-OT", "-OF", "-PF", "-nqfm", "-qfl", "-m", "flux-schnell", "-qT", "bf16", "-C"])
    assert a.config_path == "cfg.json" and a.port == 9000 and not a.offload_ae and not a.offload_text_enc and a.offload_flow
    assert a.prequantized_flow and not a.quantize_modulation and a.quantize_flow_embedder_layers and a.model_version == "flux-schnell" and a.compile


def test_optional_models_absent_offline():
    """Without local checkpoints the loaders return None instead of reaching for the network (reference util.py:262-296 downloads):
    the pipeline then takes pre-computed embeddings / returns latents."""
    import util

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16")
    assert util.load_text_encoders(cfg) == (None, None)          # clip_path is a hub id, text_enc_path is None: no local directories
    assert util.load_autoencoder(cfg) is None                    # ae_path None
    cfg.ae_path = "/nonexistent/ae.sft"
    assert util.load_autoencoder(cfg) is None


def test_autoencoder_loads_from_ae_path(tmp_path):
    """config.ae_path pointing at a real BFL-layout `ae.sft` (reference util.py:283-296): the file is read and loaded -- the round-1
    loader hit a NameError here (advisor finding), which no test reached because every test injected the state dict."""
    import torch
    from safetensors.torch import save_file

    import util
    from modules.autoencoder import AutoEncoder

    cfg = util.load_config(util.ModelVersion.flux_dev, flow_dtype="bfloat16")
    cfg.ae_params.resolution, cfg.ae_params.ch, cfg.ae_params.ch_mult, cfg.ae_params.num_res_blocks = 32, 32, [1, 2], 1
    cfg.ae_device = "cpu"
    torch.manual_seed(0)
    ref = AutoEncoder(cfg.ae_params)
    path = str(tmp_path / "ae.sft")
    save_file({k: v.contiguous() for k, v in ref.state_dict().items()}, path)
    cfg.ae_path = path
    ae = util.load_autoencoder(cfg)
    assert ae is not None and ae.encoder_loaded
    for k, v in ref.state_dict().items():
        assert torch.equal(ae.state_dict()[k].float(), v.to(torch.bfloat16).float()), k
    # decoder-only checkpoints are accepted, encoder then refuses to run
    save_file({k: v.contiguous() for k, v in ref.state_dict().items() if k.startswith("decoder.")}, path)
    assert util.load_autoencoder(cfg).encoder_loaded is False


def test_native_text_modules_keep_hf_state_dict_keys():
    """T5EncoderNative / ClipTextNative expose transformers' parameter names (so HF checkpoints load as they are), accept the tied
    T5 embedding and both CLIP key layouts, and refuse to run without a GPU."""
    import torch

    from modules.conditioner import ClipTextNative, T5EncoderNative

    t5 = T5EncoderNative(dict(vocab_size=32, d_model=64, d_kv=64, num_heads=2, d_ff=128, num_layers=2, feed_forward_proj="gated-gelu"))
    keys = set(t5.state_dict())
    assert {"shared.weight", "encoder.embed_tokens.weight", "encoder.block.0.layer.0.SelfAttention.q.weight",
            "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", "encoder.block.1.layer.1.DenseReluDense.wi_1.weight",
            "encoder.block.1.layer.1.layer_norm.weight", "encoder.final_layer_norm.weight"} <= keys
    assert "encoder.block.1.layer.0.SelfAttention.relative_attention_bias.weight" not in keys  # owned by block 0 only
    sd = {k: v.clone() for k, v in t5.state_dict().items() if k != "encoder.embed_tokens.weight"}  # tied in real checkpoints
    assert not t5.load_state_dict(sd, strict=True).missing_keys
    with pytest.raises(RuntimeError):
        t5(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(ValueError):
        T5EncoderNative(dict(vocab_size=32, d_model=64, d_kv=32, num_heads=2, d_ff=128, num_layers=1))

    clip = ClipTextNative(dict(text_config=dict(vocab_size=40, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=1)))
    ck = set(clip.state_dict())
    assert {"text_model.embeddings.token_embedding.weight", "text_model.embeddings.position_embedding.weight",
            "text_model.encoder.layers.0.self_attn.out_proj.bias", "text_model.encoder.layers.0.mlp.fc1.weight",
            "text_model.final_layer_norm.bias"} <= ck
    flat = {k[len("text_model."):]: v.clone() for k, v in clip.state_dict().items()}      # transformers 5.x layout
    flat["embeddings.position_ids"] = torch.arange(77)[None]                                # legacy buffer, ignored
    assert not clip.load_state_dict(flat, strict=True).missing_keys
    with pytest.raises(ValueError):
        ClipTextNative(dict(hidden_size=96, num_attention_heads=2, intermediate_size=64, num_hidden_layers=1, vocab_size=8))


# ---- the reference's own shipped config JSONs (tests/golden/configs = /root/reference/configs, category-(b) fixtures: the schema IS the
# interface) -------------------------------------------------------------------------------------------------------------------------
def _ref_configs():
    import glob

    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs")
    return sorted(glob.glob(os.path.join(d, "*.json")))


def test_every_reference_config_json_parses_and_resolves_dtypes():
    """All 11 JSONs of the reference load through util.load_config_from_path unchanged (unknown keys such as offload_ae / offload_text_enc
    are ignored like in the reference, util.py:38-79), every one of them asks for flow_dtype float16, and the engine dtype policy maps
    that to bf16 parameters with float16 kept as the model's I/O dtype -- no override needed (they used to raise)."""
    import warnings

    import torch
    import util

    paths = _ref_configs()
    assert len(paths) == 11
    for path in paths:
        spec = util.load_config_from_path(path)
        assert spec.flow_dtype == "float16" and spec.params.hidden_size == 3072
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert util.engine_flow_dtype(spec.flow_dtype) == torch.bfloat16
        # the model tree of this config on the meta device: parameters in bf16, I/O dtype float16, F8Linear placement per the config's flags
        spec.params.depth, spec.params.depth_single_blocks = 1, 1  # (structure check only; the GPU test loads + runs every config)
        spec.ckpt_path = None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = util.load_flow_model(spec)
        assert m.dtype == torch.float16
        from float8_quantize import F8Linear

        n_f8 = sum(isinstance(x, F8Linear) for x in m.modules())
        if spec.prequantized_flow:
            assert n_f8 > 0 and isinstance(m.double_blocks[0].img_attn.qkv, F8Linear)
        else:
            assert n_f8 == 0 and m.double_blocks[0].img_attn.qkv.weight.dtype == torch.bfloat16


def test_list_prompt_sizes_the_batch():
    """reference flux_pipeline.py:267-278: a list prompt with one noise sample -> batch = len(list); dict conditioning still broadcasts."""
    import torch
    import flux_pipeline as fp

    pipe = fp.FluxPipeline.__new__(fp.FluxPipeline)
    pipe.device_flux, pipe.dtype, pipe.t5, pipe.clip, pipe.debug = torch.device("cpu"), torch.bfloat16, None, None, False
    pipe.device_clip = torch.device("cpu")
    img = torch.randn(1, 16, 8, 8)
    tok, ids, vec, txt, tids = pipe.prepare(img, {"txt": torch.zeros(1, 4, 8), "vec": torch.zeros(1, 6)})
    assert tok.shape == (1, 16, 64) and txt.shape[0] == 1
    with pytest.raises(RuntimeError, match="no text encoders"):
        pipe.prepare(img, ["a cat", "a dog"])
    with pytest.raises(TypeError):
        pipe.prepare(img, ["a cat", 3])

    class FakeEnc:
        pass

    pipe.t5 = pipe.clip = FakeEnc()
    import flux_emphasis

    calls = []

    def fake(pipe_, prompt, num_images_per_prompt=1, **kw):
        calls.append((prompt, num_images_per_prompt))
        n = num_images_per_prompt
        return torch.full((n, 6), float(len(prompt))), torch.full((n, 4, 8), float(len(prompt))), torch.zeros(n, 4, 3)

    old = flux_emphasis.get_weighted_text_embeddings_flux
    flux_emphasis.get_weighted_text_embeddings_flux = fake
    try:
        tok, ids, vec, txt, tids = pipe.prepare(img, ["a cat", "a big dog"])
        assert tok.shape == (2, 16, 64) and ids.shape[0] == 2 and vec.shape == (2, 6) and txt.shape == (2, 4, 8) and tids.shape == (2, 4, 3)
        assert torch.equal(tok[0], tok[1]) and vec[0, 0] == 5 and vec[1, 0] == 9 and calls == [("a cat", 1), ("a big dog", 1)]
        tok, ids, vec, txt, tids = pipe.prepare(torch.randn(3, 16, 8, 8), "one prompt")
        assert calls[-1] == ("one prompt", 3) and vec.shape[0] == 3
    finally:
        flux_emphasis.get_weighted_text_embeddings_flux = old


def test_diffusers_lora_key_conversion_matches_reference_fixture():
    """lora_loading.convert_diffusers_to_flux_transformer_checkpoint / resolve_lora_state_dict against the outputs of the UNMODIFIED reference
    functions (lora_loading.py:62-432,580-606) on synthetic diffusers-format LoRA dicts (oracle/gen_golden_lora_diffusers.py): same key set,
    same tensors, same row order of the fused qkv / linear1 concatenations, zero-filled missing members, untouched `.alpha` leftovers, and the
    guidance embedder left unconverted for flux-schnell.  Missing single-block members raise KeyError like the reference."""
    import torch
    from safetensors.torch import load_file

    import lora_loading as ll

    fx = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_lora_diffusers.safetensors"))
    for name, has_guidance, src in (("full", True, "full"), ("sparse", True, "sparse"), ("noguidance", False, "full")):
        inp = {k[len(src) + 4:]: v for k, v in fx.items() if k.startswith(src + ":in:")}
        want = {k[len(name) + 5:]: v for k, v in fx.items() if k.startswith(name + ":out:")}
        assert inp and want
        keys, got = ll.resolve_lora_state_dict({k: v.clone() for k, v in inp.items()}, has_guidance=has_guidance)
        assert set(got) == set(want), (name, sorted(set(got) ^ set(want))[:6])
        for k in want:
            assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (name, k)
        if name != "noguidance":
            assert len(keys) == int(fx[f"{name}:n_keys_without_ab"])
        # what apply_lora_to_model will see: fused qkv in the uneven-rank form, linear1 with four chunks
        a, b = got["double_blocks.0.img_attn.qkv.lora_A.weight"], got["double_blocks.0.img_attn.qkv.lora_B.weight"]
        assert a.shape[0] == 3 * b.shape[1] and got["single_blocks.0.linear1.lora_A.weight"].shape[0] == 4 * got["single_blocks.0.linear1.lora_B.weight"].shape[1]
    broken = {k[8:]: v for k, v in fx.items() if k.startswith("full:in:") and "single_transformer_blocks.3.attn.to_k" not in k}
    with pytest.raises(KeyError):
        ll.resolve_lora_state_dict(broken, has_guidance=True)
    # kohya files keep working through the same entry point
    _, k2 = ll.resolve_lora_state_dict({"lora_unet_double_blocks_0_img_attn_qkv.lora_down.weight": torch.zeros(2, 4),
                                        "lora_unet_single_blocks_1_linear2.lora_up.weight": torch.zeros(4, 2), "not_a_lo_ra_key": torch.zeros(1)})
    assert set(k2) == {"double_blocks.0.img_attn.qkv.lora_A.weight", "single_blocks.1.linear2.lora_B.weight"}


def test_rocprof_summary_reads_csv_kernel_trace(tmp_path):
    """tools/rocprof_summary.py turns a `rocprofv3 --kernel-trace --output-format csv` trace into the per-kernel table committed under
    profiles/; --steady keeps the fused steps only (dispatches after the last calibration kernel, from the first euler_kernel on)."""
    import subprocess
    import sys

    rows = [("amax_kernel", 0, 10), ("calib_update_kernel", 10, 20), ("gemm_pp_kernel<true>", 20, 120), ("euler_kernel", 120, 125)]
    t = 125
    for _ in range(3):  # three graph-replayed steps: two GEMMs + one attention + euler each
        for name, dur in (("gemm_pp_kernel<true>", 100), ("gemm_pp_kernel<true>", 100), ("attention2_kernel<1, true, false>", 50), ("euler_kernel", 5)):
            rows.append((name, t, t + dur))
            t += dur
    d = tmp_path / "prof"
    d.mkdir()
    with open(d / "x_kernel_trace.csv", "w") as f:
        f.write('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for name, s, e in rows:
            f.write(f'"KERNEL_DISPATCH","void (anonymous namespace)::{name}(Args)",{s},{e}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rocprof_summary.py"), str(d), "--steady"], capture_output=True, text=True, check=True).stdout
    assert "3 graph-replayed denoise steps of the last request" in out and "4 launches/step" in out, out
    line = [ln for ln in out.splitlines() if ln.startswith("gemm_pp_kernel<true>")][0].split()
    assert line[-4:-1] == ["6", "0.6", "0.10"], line  # 6 calls, 600 ns = 0.6 us in total, 0.10 us each
    allk = subprocess.run([sys.executable, os.path.join(root, "tools", "rocprof_summary.py"), str(d)], capture_output=True, text=True, check=True).stdout
    assert [ln for ln in allk.splitlines() if ln.startswith("gemm_pp_kernel<true>")][0].split()[-4] == "7"


def test_bench_power_sampler_reads_the_timed_devices_sensor(tmp_path):
    """bench.py's board-power sampler must read the hwmon directory of the PCI device the work runs on: on a node whose other GPUs are
    hidden from the process, card0 is usually another GPU (round 4: 245 / 320 / 635 W from neighbours beside 1258 - 1277 W).  Fake sysfs:
    two devices, card0 = the wrong one; without a match the summary says so."""
    import importlib.util
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sysfs = tmp_path / "sys"
    for i, (bdf, uw) in enumerate((("0000:05:00.0", 245_000_000), ("0000:c1:00.0", 1_270_000_000))):
        hw = sysfs / "bus" / "pci" / "devices" / bdf / "hwmon" / f"hwmon{i + 3}"
        hw.mkdir(parents=True)
        (hw / "power1_input").write_text(str(uw))
        (hw / "freq1_input").write_text("2000000000")
        card = sysfs / "class" / "drm" / f"card{i}"
        card.mkdir(parents=True)
        os.symlink(sysfs / "bus" / "pci" / "devices" / bdf, card / "device")

    def sample(bdf):
        ps = bench._PowerSampler(bdf, sysfs_root=str(sysfs))
        ps.start()
        time.sleep(0.1)
        ps.stop()
        return ps.summary()

    right = sample("0000:C1:00.0")
    assert right["device_matched"] and abs(right["watts_mean"] - 1270.0) < 1e-6 and "0000:c1:00.0" in right["source"] and "note" not in right
    unknown = sample(None)
    assert unknown["device_matched"] is False and "note" in unknown  # first hwmon found (card0 = the neighbour), flagged
    assert abs(unknown["watts_mean"] - 245.0) < 1e-6
    assert bench._PowerSampler("0000:ff:00.0", sysfs_root=str(tmp_path / "nothing")).summary() is None
