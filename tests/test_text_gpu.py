"""SURVEY.md §8(f) row 2 on the GPU: the native T5 / CLIP text encoders (modules/conditioner.py over libfluxmi) and the whole
conditioning path (flux_emphasis.py -> HFEmbedder -> FluxPipeline.prepare / generate) against the fixtures written by
oracle/gen_golden_text.py: transformers' fp32 outputs for the encoders, the UNMODIFIED reference's outputs for the weighted prompts.

Tolerance: bf16 execution is judged by its distance to the fp32 result, with transformers' own bf16 run as the yardstick (x1.5) --
the same bar as the VAE (tests/test_engine_gpu.py::test_vae_decoder_matches_reference_fixture)."""
import io
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T5_CFG = dict(vocab_size=0, d_model=128, d_kv=64, num_heads=4, d_ff=256, num_layers=2, feed_forward_proj="gated-gelu",
              relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)
CLIP_CFG = dict(vocab_size=0, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2, max_position_embeddings=77,
                hidden_act="quick_gelu", layer_norm_eps=1e-5)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def gold():
    from safetensors.torch import load_file

    return load_file(os.path.join(GOLD, "g9_text.safetensors"))


@pytest.fixture(scope="module")
def embedders(gold, dev):
    from transformers import CLIPTokenizer, T5Tokenizer

    from modules.conditioner import HFEmbedder

    clip_tok, t5_tok = CLIPTokenizer.from_pretrained(os.path.join(GOLD, "tok_clip")), T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok_t5"))
    t5_sd = {k[3:]: v for k, v in gold.items() if k.startswith("t5.")}
    clip_sd = {k[5:]: v for k, v in gold.items() if k.startswith("clip.")}
    t5_kw = dict(hf_config=dict(T5_CFG, vocab_size=t5_sd["shared.weight"].shape[0]), state_dict=t5_sd, tokenizer=t5_tok)
    clip_kw = dict(hf_config=dict(CLIP_CFG, vocab_size=clip_sd["embeddings.token_embedding.weight"].shape[0], eos_token_id=clip_tok.eos_token_id),
                   state_dict=clip_sd, tokenizer=clip_tok)
    clip = HFEmbedder("clip-fixture", max_length=77, device=dev, is_clip=True, **clip_kw)
    t5 = HFEmbedder("t5-fixture", max_length=512, device=dev, **t5_kw)
    return clip, t5, clip_kw, t5_kw


def test_native_t5_encoder_matches_transformers_fixture(gold, embedders, dev):
    _, t5, _, _ = embedders
    out = t5.hf_module(gold["ids_t5"].to(dev), attention_mask=None, output_hidden_states=False)["last_hidden_state"]
    assert out.shape == gold["hf_t5_fp32"].shape and out.dtype == torch.bfloat16 and torch.isfinite(out).all()
    e_native, e_ref = rel(out, gold["hf_t5_fp32"]), rel(gold["hf_t5_bf16"], gold["hf_t5_fp32"])
    print(f"T5 encoder: native vs fp32 {e_native:.3e}; transformers bf16 vs fp32 {e_ref:.3e}")
    assert e_native <= 1.5 * e_ref
    with pytest.raises(NotImplementedError):
        t5.hf_module(gold["ids_t5"].to(dev), attention_mask=torch.ones_like(gold["ids_t5"]))


def test_native_clip_text_matches_transformers_fixture(gold, embedders, dev):
    clip, _, _, _ = embedders
    out = clip.hf_module(gold["ids_clip"].to(dev), attention_mask=None, output_hidden_states=True)
    e_ref = rel(gold["hf_clip_pooled_bf16"], gold["hf_clip_pooled_fp32"])
    e_h, e_p = rel(out["last_hidden_state"], gold["hf_clip_hidden_fp32"]), rel(out["pooler_output"], gold["hf_clip_pooled_fp32"])
    print(f"CLIP text: native hidden vs fp32 {e_h:.3e}, pooled {e_p:.3e}; transformers bf16 pooled vs fp32 {e_ref:.3e}")
    assert out["pooler_output"].shape == gold["hf_clip_pooled_fp32"].shape
    assert e_p <= 1.5 * e_ref and e_h <= 1.5 * e_ref


def test_hfembedder_forward_and_weighted_embeddings_match_reference(gold, embedders, dev):
    """HFEmbedder.forward (conditioner.py:102-117) and get_weighted_text_embeddings_flux (flux_emphasis.py:307-447) through the native
    encoders vs the unmodified reference's fp32 outputs for the same prompts / tokenizers / weights."""
    import types

    import flux_emphasis as fe

    clip, t5, _, _ = embedders
    meta = json.load(open(os.path.join(GOLD, "g9_text.json")))
    pipe = types.SimpleNamespace(name="flux-dev", clip=clip, t5=t5)
    e_t5 = rel(gold["hf_t5_bf16"], gold["hf_t5_fp32"])
    e_clip = rel(gold["hf_clip_pooled_bf16"], gold["hf_clip_pooled_fp32"])
    for i, prompt in enumerate(meta["prompts"]):
        vec, txt, ids = fe.get_weighted_text_embeddings_flux(pipe, prompt, num_images_per_prompt=2, device=dev, target_device=dev,
                                                             target_dtype=torch.bfloat16)
        assert vec.shape == (2, 128) and txt.shape == (2, 512, 128) and ids.shape == (2, 512, 3) and vec.dtype == torch.bfloat16
        ev, et = rel(vec[:1], gold[f"emph{i}.vec"]), rel(txt[:1], gold[f"emph{i}.txt"])
        print(f"prompt {i}: vec {ev:.3e} (yardstick {e_clip:.3e})  txt {et:.3e} (yardstick {e_t5:.3e})")
        assert ev <= 2.0 * e_clip and et <= 2.0 * e_t5, prompt
    # plain forward: CLIP -> pooled [B, D], T5 -> hidden states [B, max_length, D]
    assert clip(["a photo of a cat", "sky"]).shape == (2, 128)
    assert t5(["a photo of a cat"]).shape == (1, 512, 128)


def test_pipeline_generate_from_a_prompt_string(embedders, dev):
    """generate('a (red:1.5) cat ...') end to end: tokenizers -> native CLIP / T5 -> prompt weighting -> denoise loop -> latents, and the
    same call with the embeddings it computed handed in as a dict gives the same latents."""
    from test_engine_gpu import tiny_config

    from flux_pipeline import FluxPipeline
    from fluxmi import synth

    _, _, clip_kw, t5_kw = embedders
    cfg = tiny_config()
    cfg.params.vec_in_dim = 128
    cfg.text_enc_device = str(dev)
    pipe = FluxPipeline.load_pipeline_from_config(cfg, state_dict=synth.make_state_dict(cfg.params, seed=0), clip_kwargs=clip_kw, t5_kwargs=t5_kw)
    assert pipe.clip is not None and pipe.t5 is not None and pipe.t5.max_length == 512
    prompt = "a (red:1.5) cat on a [hill], (sky)"
    noise = pipe.get_noise(1, 64, 64, generator=torch.Generator(device=dev).manual_seed(5))
    _, _, vec, txt, txt_ids = pipe.prepare(noise, prompt)
    assert vec.shape == (1, 128) and txt.shape == (1, 512, 128) and txt_ids.shape == (1, 512, 3)
    pipe.compile(prompt={"txt": txt, "vec": vec})
    a = pipe.generate(prompt, width=64, height=64, num_steps=4, seed=11, silent=True)
    b = pipe.generate({"txt": txt, "vec": vec}, width=64, height=64, num_steps=4, seed=11, silent=True)
    assert a.shape == (1, 16, 8, 8) and torch.isfinite(a).all() and torch.equal(a, b)
    # a list prompt with one noise sample sizes the batch (reference flux_pipeline.py:267-278); every prompt is embedded on its own
    tok2, ids2, vec2, txt2, tids2 = pipe.prepare(noise, [prompt, "a dog"])
    assert tok2.shape[0] == 2 and vec2.shape == (2, 128) and txt2.shape == (2, 512, 128) and torch.equal(tok2[0], tok2[1])
    assert torch.equal(vec2[0], vec[0]) and torch.equal(txt2[0], txt[0]) and not torch.equal(txt2[1], txt2[0])
    c = pipe.generate([prompt, "a dog"], width=64, height=64, num_steps=4, seed=11, silent=True)
    assert c.shape == (2, 16, 8, 8) and torch.equal(c[0], a[0])
    with pytest.raises(TypeError):
        pipe.prepare(noise, ["a", 3])


def test_full_width_text_encoders_match_oracle(dev):
    """T5-v1.1-XXL and CLIP-L layer shapes (d_model 4096 / 64 heads / d_ff 10240, L = 512; hidden 768 / 12 heads / 3072, L = 77), two
    layers each with random weights: native vs the fp32 oracle run on the GPU, yardstick = the oracle's own bf16 run (torch bf16 ops,
    i.e. what the reference executes)."""
    import text_oracle as to

    from modules.conditioner import ClipTextNative, T5EncoderNative

    g = torch.Generator().manual_seed(7)

    def fill(module, gain):
        sd = {}
        for k, v in module.state_dict().items():
            if "norm" in k and k.endswith("weight"):
                t = 1 + 0.1 * torch.randn(v.shape, generator=g)
            elif k.endswith("bias"):
                t = 0.05 * torch.randn(v.shape, generator=g)
            elif "embed" in k or k == "shared.weight" or "relative_attention_bias" in k:
                t = torch.randn(v.shape, generator=g)
            else:
                t = torch.randn(v.shape, generator=g) * (gain / v.shape[-1] ** 0.5)
            sd[k] = t.to(torch.bfloat16)
        return sd

    t5 = T5EncoderNative(dict(vocab_size=1000, d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=2, feed_forward_proj="gated-gelu"))
    sd = fill(t5, 0.8)
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    t5.load_state_dict(sd)
    t5.to(dev, dtype=torch.bfloat16)
    ids = torch.randint(0, 1000, (1, 512), generator=g).to(dev)
    out = t5(ids)["last_hidden_state"]
    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    cfg = dict(num_layers=2, num_heads=64, d_kv=64, eps=1e-6)
    ref32 = to.t5_encoder(sd_dev, cfg, ids, torch.float32)
    ref16 = to.t5_encoder(sd_dev, cfg, ids, torch.bfloat16)
    e_native, e_ref = rel(out, ref32), rel(ref16, ref32)
    print(f"T5-XXL width: native vs fp32 oracle {e_native:.3e}; oracle bf16 vs fp32 {e_ref:.3e}")
    assert out.shape == (1, 512, 4096) and e_native <= 1.5 * e_ref

    clip = ClipTextNative(dict(vocab_size=1000, hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=2,
                               max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999))
    sd = fill(clip, 1.0)
    clip.load_state_dict(sd)
    clip.to(dev, dtype=torch.bfloat16)
    ids = torch.randint(0, 998, (2, 77), generator=g)
    ids[0, 20:] = 999
    ids[1, 76] = 999
    ids = ids.to(dev)
    out = clip(ids)
    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    cfg = dict(num_layers=2, num_heads=12, eps=1e-5, eos_token_id=999)
    h32, p32 = to.clip_text(sd_dev, cfg, ids, torch.float32)
    h16, p16 = to.clip_text(sd_dev, cfg, ids, torch.bfloat16)
    print(f"CLIP-L width: native pooled vs fp32 oracle {rel(out['pooler_output'], p32):.3e}, hidden {rel(out['last_hidden_state'], h32):.3e}; "
          f"oracle bf16 vs fp32 pooled {rel(p16, p32):.3e}, hidden {rel(h16, h32):.3e}")
    assert rel(out["pooler_output"], p32) <= 1.5 * rel(p16, p32) and rel(out["last_hidden_state"], h32) <= 1.5 * rel(h16, h32)
