"""SURVEY.md §8(f) row 2 on CPU: the prompt-weighting host logic (flux_emphasis.py mirror) against what the UNMODIFIED reference module
returned (tests/golden/g9_text.json / g9_text.safetensors, written by oracle/gen_golden_text.py), and the text-encoder restatements
of oracle/text_oracle.py against the transformers outputs stored in the same fixture."""
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLD, "g9_text.json")))


@pytest.fixture(scope="module")
def gold():
    from safetensors.torch import load_file

    return load_file(os.path.join(GOLD, "g9_text.safetensors"))


@pytest.fixture(scope="module")
def toks():
    from transformers import CLIPTokenizer, T5Tokenizer

    return CLIPTokenizer.from_pretrained(os.path.join(GOLD, "tok_clip")), T5Tokenizer.from_pretrained(os.path.join(GOLD, "tok_t5"))


def test_parse_prompt_attention_matches_reference(meta):
    import flux_emphasis as fe

    assert len(meta["parse"]) >= 12
    for text, want in meta["parse"].items():
        assert fe.parse_prompt_attention(text) == want, text
    # 400 seeded random strings over the grammar's alphabet, parsed by the unmodified reference when the fixture was written
    assert len(meta["fuzz"]) == 400
    for text, want in meta["fuzz"]:
        if want == "ValueError":
            with pytest.raises(ValueError):
                fe.parse_prompt_attention(text)
        else:
            assert fe.parse_prompt_attention(text) == want, repr(text)
    # the reference's own doctest vectors (flux_emphasis.py:27-47)
    assert fe.parse_prompt_attention("normal text") == [["normal text", 1.0]]
    assert fe.parse_prompt_attention("an (important) word") == [["an ", 1.0], ["important", 1.1], [" word", 1.0]]
    assert fe.parse_prompt_attention("(unbalanced") == [["unbalanced", 1.1]]
    assert fe.parse_prompt_attention("\\(literal\\]") == [["(literal]", 1.0]]
    assert fe.parse_prompt_attention("(unnecessary)(parens)") == [["unnecessaryparens", 1.1]]
    got = fe.parse_prompt_attention("a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).")
    want = [["a ", 1.0], ["house", 1.5730000000000004], [" ", 1.1], ["on", 1.0], [" a ", 1.1], ["hill", 0.55], [", sun, ", 1.1],
            ["sky", 1.4641000000000006], [".", 1.1]]
    assert got == want
    # pieces the fixtures do not reach: escapes, stray closers, weights without an opener, a malformed number
    assert fe.parse_prompt_attention("a \\\\ b \\ c") == [["a \\ b  c", 1.0]]
    assert fe.parse_prompt_attention("x) y] :1.5) z") == [["x) y] :1.5) z", 1.0]]
    assert fe.parse_prompt_attention("(a BREAK b)") == [["a", 1.1], ["BREAK", -1.1], ["b", 1.1]]
    with pytest.raises(ValueError):
        fe.parse_prompt_attention("(a:1.2.3)")


def test_tokens_and_groups_match_reference(meta, toks):
    import flux_emphasis as fe

    clip_tok, t5_tok = toks
    for prompt in meta["prompts"]:
        tk, tw = fe.get_prompts_tokens_with_weights(clip_tok, prompt)
        t5k, t5w = fe.get_prompts_tokens_with_weights(t5_tok, prompt)
        assert [tk, tw] == meta["tokens"][prompt]["clip"], prompt
        assert [t5k, t5w] == meta["tokens"][prompt]["t5"], prompt
        g = fe.group_tokens_and_weights(list(tk), list(tw), pad_last_block=True, bos=clip_tok.bos_token_id, eos=clip_tok.eos_token_id, max_length=77)
        g5 = fe.group_tokens_and_weights(list(t5k), list(t5w), pad_last_block=True, bos=None, eos=t5_tok.eos_token_id, max_length=512, pad_tokens=False)
        assert [list(g[0]), list(g[1])] == meta["groups"][prompt]["clip"], prompt
        assert [list(g5[0]), list(g5[1])] == meta["groups"][prompt]["t5"], prompt
    ids, w = meta["long_group"]["in"]
    got = fe.group_tokens_and_weights(list(ids), list(w), pad_last_block=True, bos=1000, eos=1001, max_length=77)
    assert [list(got[0]), list(got[1])] == meta["long_group"]["clip"]
    a, b = list(ids), list(w)
    got = fe.group_tokens_and_weights(a, b, pad_last_block=True, bos=None, eos=1, max_length=64, pad_tokens=False)
    assert [list(got[0]), list(got[1])] == meta["long_group"]["t5"]
    assert len(a) == len(b) == 170 % 62  # like the reference, the consumed tokens are gone from the caller's lists


def test_apply_weights_bit_equal_to_reference(gold):
    import flux_emphasis as fe
    import text_oracle as to

    out = fe.apply_weights(gold["aw.tokens"], gold["aw.weights"], gold["aw.emb"].clone(), 1)
    assert torch.equal(out, gold["aw.out"])
    assert torch.equal(to.apply_weights(gold["aw.tokens"], gold["aw.weights"], gold["aw.emb"], 1), gold["aw.out"])
    x = gold["aw.emb"]
    assert torch.equal(fe.standardize_tensor(x, 0.25, 2.0), (x - x.mean()) / x.std() * 2.0 + 0.25)


def test_text_oracle_matches_transformers_fixture(gold):
    """oracle/text_oracle.py (published T5 v1.1 encoder / CLIP text algorithms) vs transformers 5.15.0 outputs on the fixture's tiny
    random models: fp32 to reduction-order noise; the bf16 run no further from fp32 than transformers' own bf16 run (x1.5)."""
    import text_oracle as to

    t5_sd = {k[3:]: v.float() for k, v in gold.items() if k.startswith("t5.")}
    clip_sd = {k[5:]: v.float() for k, v in gold.items() if k.startswith("clip.")}
    t5_cfg = dict(num_layers=2, num_heads=4, d_kv=64, eps=1e-6)
    clip_cfg = dict(num_layers=2, num_heads=2, eps=1e-5, eos_token_id=int(gold["ids_clip"][0, -1]))
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    with torch.no_grad():
        o = to.t5_encoder(t5_sd, t5_cfg, gold["ids_t5"])
        h, p = to.clip_text(clip_sd, clip_cfg, gold["ids_clip"])
        ob = to.t5_encoder(t5_sd, t5_cfg, gold["ids_t5"], torch.bfloat16)
        _, pb = to.clip_text(clip_sd, clip_cfg, gold["ids_clip"], torch.bfloat16)
    assert rel(o, gold["hf_t5_fp32"]) < 2e-6
    assert rel(h, gold["hf_clip_hidden_fp32"]) < 2e-6 and rel(p, gold["hf_clip_pooled_fp32"]) < 2e-6
    assert rel(ob, gold["hf_t5_fp32"]) <= 1.5 * rel(gold["hf_t5_bf16"], gold["hf_t5_fp32"])
    assert rel(pb, gold["hf_clip_pooled_fp32"]) <= 1.5 * rel(gold["hf_clip_pooled_bf16"], gold["hf_clip_pooled_fp32"])
    # relative-position bucketing: values of transformers' T5Attention._relative_position_bucket(bidirectional, 32, 128)
    b = to.t5_relative_position_bucket(torch.tensor([0, 1, -1, 7, 8, -8, 15, 16, 127, 128, 500, -500]))
    assert b.tolist() == [0, 17, 1, 23, 24, 8, 25, 26, 31, 31, 31, 15]


def test_text_pipeline_oracle_matches_reference_embeddings(meta, gold, toks):
    """The whole conditioning path restated on CPU -- flux_emphasis mirror driving the ORACLE encoders -- vs what the unmodified
    reference returned through transformers' models (fp32): vec / txt of every fixture prompt."""
    import types

    import flux_emphasis as fe
    import text_oracle as to

    clip_tok, t5_tok = toks
    t5_sd = {k[3:]: v.float() for k, v in gold.items() if k.startswith("t5.")}
    clip_sd = {k[5:]: v.float() for k, v in gold.items() if k.startswith("clip.")}
    t5_cfg = dict(num_layers=2, num_heads=4, d_kv=64, eps=1e-6)
    clip_cfg = dict(num_layers=2, num_heads=2, eps=1e-5, eos_token_id=clip_tok.eos_token_id)

    def t5(ids, **_):
        return {"last_hidden_state": to.t5_encoder(t5_sd, t5_cfg, ids)}

    def clip(ids, **_):
        h, p = to.clip_text(clip_sd, clip_cfg, ids)
        return {"last_hidden_state": h, "pooler_output": p}

    pipe = types.SimpleNamespace(name="flux-dev", clip=types.SimpleNamespace(tokenizer=clip_tok, hf_module=clip),
                                 t5=types.SimpleNamespace(tokenizer=t5_tok, hf_module=t5))
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    for i, prompt in enumerate(meta["prompts"]):
        vec, txt, ids = fe.get_weighted_text_embeddings_flux(pipe, prompt, num_images_per_prompt=2, device=torch.device("cpu"),
                                                             target_device=torch.device("cpu"), target_dtype=torch.float32)
        assert vec.shape == (2, 128) and txt.shape == (2, 512, 128) and ids.shape == (2, 512, 3) and not ids.any()
        assert torch.equal(vec[0], vec[1]) and torch.equal(txt[0], txt[1])
        assert rel(vec[:1], gold[f"emph{i}.vec"]) < 5e-6, prompt
        assert rel(txt[:1], gold[f"emph{i}.txt"]) < 5e-6, prompt
