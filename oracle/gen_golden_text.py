"""Generates tests/golden/g9_text.safetensors, g9_text.json and the tiny tokenizers under tests/golden/tok_t5, tok_clip.

Pins oracle/text_oracle.py (T5 encoder + CLIP text model restatements) against `transformers` (the reference's dependency for this
path, modules/conditioner.py:74-93) on tiny random models, and records what the UNMODIFIED reference flux_emphasis.py
(/root/reference, imported on CPU) returns for a set of weighted prompts through those models.
    python oracle/gen_golden_text.py        (needs /root/reference; the committed fixtures are what the tests read)"""
import json
import os
import shutil
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import flux_emphasis as ref_emph  # noqa: E402  (the reference's)
import transformers  # noqa: E402
from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer, T5Config, T5EncoderModel, T5Tokenizer  # noqa: E402

import text_oracle as to  # noqa: E402

CORPUS = ["a photo of a cat on a hill", "the house on the hill under a blue sky", "sun, sky and sea. a red car",
          "an astronaut riding a horse on mars, highly detailed", "a painting of a small village by the river at night"]
PROMPTS = ["a photo of a cat", "a (red:1.5) cat on a [hill], (sky)", "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).",
           "an astronaut riding a (horse:0.8) on mars BREAK highly detailed", ""]
PARSE_CASES = ["normal text", "an (important) word", "(unbalanced", "\\(literal\\]", "(unnecessary)(parens)",
               "a (((house:1.3)) [on] a (hill:0.5), sun, (((sky))).", "x BREAK y", "", "a [b] (c:2) d:", "((a):1.2)"]
T5_CFG = dict(d_model=128, d_kv=64, num_heads=4, d_ff=256, num_layers=2)
CLIP_CFG = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2)


def build_tokenizers():
    import sentencepiece as spm
    def bytes_to_unicode():  # the GPT-2 / CLIP byte <-> printable-character table
        bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\u00a1"), ord("\u00ac") + 1)) + list(range(ord("\u00ae"), ord("\u00ff") + 1))
        cs, n = bs[:], 0
        for b in range(256):
            if b not in bs:
                bs.append(b)
                cs.append(256 + n)
                n += 1
        return dict(zip(bs, [chr(c) for c in cs]))

    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "corpus.txt"), "w") as f:
        f.write("\n".join(CORPUS * 20))
    spm.SentencePieceTrainer.train(input=os.path.join(tmp, "corpus.txt"), model_prefix=os.path.join(tmp, "spiece"), vocab_size=64,
                                   model_type="unigram", pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False)
    t5_dir = os.path.join(GOLD, "tok_t5")
    shutil.rmtree(t5_dir, ignore_errors=True)
    os.makedirs(os.path.join(tmp, "t5"))
    shutil.copy(os.path.join(tmp, "spiece.model"), os.path.join(tmp, "t5", "spiece.model"))
    T5Tokenizer.from_pretrained(os.path.join(tmp, "t5")).save_pretrained(t5_dir)  # tokenizer.json (text) instead of the binary model
    # the sentencepiece nmt_nfkc character map is 320 KB of base64; this fixture only sees ASCII prompts -> plain NFKC keeps it small
    tj = json.load(open(os.path.join(t5_dir, "tokenizer.json")))
    tj["normalizer"] = {"type": "NFKC"}
    json.dump(tj, open(os.path.join(t5_dir, "tokenizer.json"), "w"))
    # CLIP: byte-level BPE with a handful of merges
    base = list(bytes_to_unicode().values())
    vocab = {}
    for c in base:
        vocab[c] = len(vocab)
    for c in base:
        vocab[c + "</w>"] = len(vocab)
    merges = ["c a", "ca t</w>", "h i", "hi l", "hil l</w>", "s k", "sk y</w>", "o n</w>", "r e", "re d</w>", "h o", "ho r", "hor s", "hors e</w>"]
    for m in merges:
        a, b = m.split()
        vocab[a + b] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    clip_dir = os.path.join(GOLD, "tok_clip")
    shutil.rmtree(clip_dir, ignore_errors=True)
    os.makedirs(os.path.join(tmp, "clip"))
    json.dump(vocab, open(os.path.join(tmp, "clip", "vocab.json"), "w"))
    open(os.path.join(tmp, "clip", "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(merges))
    CLIPTokenizer(os.path.join(tmp, "clip", "vocab.json"), os.path.join(tmp, "clip", "merges.txt"), model_max_length=77).save_pretrained(clip_dir)
    return T5Tokenizer.from_pretrained(t5_dir), CLIPTokenizer.from_pretrained(clip_dir)


def randomize(model, seed, gain=1.5):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias") and "relative" not in n:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn(p.shape, generator=g))
            elif "embed" in n or "shared" in n:
                p.copy_(torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (gain / p.shape[-1] ** 0.5))
            p.copy_(p.to(torch.bfloat16).float())  # text_enc_dtype = bfloat16 in the reference configs
    return model.eval()


def main():
    from safetensors.torch import save_file

    t5_tok, clip_tok = build_tokenizers()
    t5 = randomize(T5EncoderModel(T5Config(vocab_size=len(t5_tok), feed_forward_proj="gated-gelu", relative_attention_num_buckets=32,
                                           relative_attention_max_distance=128, **T5_CFG)), 1, gain=0.8)
    clip = randomize(CLIPTextModel(CLIPTextConfig(vocab_size=len(clip_tok), max_position_embeddings=77, hidden_act="quick_gelu",
                                                  bos_token_id=clip_tok.bos_token_id, eos_token_id=clip_tok.eos_token_id,
                                                  pad_token_id=clip_tok.pad_token_id, **CLIP_CFG)), 2)
    t5_sd = {k: v.clone() for k, v in t5.state_dict().items()}
    clip_sd = {k: v.clone() for k, v in clip.state_dict().items()}
    t5_cfg = dict(num_layers=T5_CFG["num_layers"], num_heads=T5_CFG["num_heads"], d_kv=T5_CFG["d_kv"], eps=1e-6)
    clip_cfg = dict(num_layers=CLIP_CFG["num_hidden_layers"], num_heads=CLIP_CFG["num_attention_heads"], eps=1e-5,
                    eos_token_id=clip_tok.eos_token_id)
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()

    # ---- the encoders alone: HF fp32 vs the restatement, HF bf16 as the error yardstick ------------------------------------------
    g = torch.Generator().manual_seed(3)
    ids_t5 = torch.randint(0, len(t5_tok), (2, 40), generator=g)
    ids_clip = torch.randint(0, len(clip_tok) - 2, (2, 77), generator=g)
    ids_clip[:, 0] = clip_tok.bos_token_id
    ids_clip[0, 9:] = clip_tok.eos_token_id
    ids_clip[1, 30:] = clip_tok.eos_token_id
    with torch.no_grad():
        hf_t5 = t5(input_ids=ids_t5, attention_mask=None).last_hidden_state
        o_t5 = to.t5_encoder(t5_sd, t5_cfg, ids_t5)
        hf_clip = clip(input_ids=ids_clip, attention_mask=None)
        o_clip_h, o_clip_p = to.clip_text(clip_sd, clip_cfg, ids_clip)
        hf_t5_bf = t5.to(torch.bfloat16)(input_ids=ids_t5, attention_mask=None).last_hidden_state.float()
        hf_clip_bf = clip.to(torch.bfloat16)(input_ids=ids_clip, attention_mask=None)
        t5.float(), clip.float()
        o_t5_bf = to.t5_encoder(t5_sd, t5_cfg, ids_t5, torch.bfloat16).float()
        o_clip_bf = to.clip_text(clip_sd, clip_cfg, ids_clip, torch.bfloat16)
    print(f"transformers {transformers.__version__}")
    print(f"T5   fp32: restatement vs HF rel-L2 {rel(o_t5, hf_t5):.2e} (max abs {(o_t5 - hf_t5).abs().max():.2e}); bf16 vs fp32: HF {rel(hf_t5_bf, hf_t5):.2e}, "
          f"restatement {rel(o_t5_bf, hf_t5):.2e}")
    print(f"CLIP fp32: restatement vs HF hidden {rel(o_clip_h, hf_clip.last_hidden_state):.2e}, pooled {rel(o_clip_p, hf_clip.pooler_output):.2e}; bf16 vs "
          f"fp32 pooled: HF {rel(hf_clip_bf.pooler_output, hf_clip.pooler_output):.2e}, restatement {rel(o_clip_bf[1], hf_clip.pooler_output):.2e}")
    assert rel(o_t5, hf_t5) < 2e-6 and rel(o_clip_h, hf_clip.last_hidden_state) < 2e-6 and rel(o_clip_p, hf_clip.pooler_output) < 2e-6
    assert rel(o_t5_bf, hf_t5) <= 1.5 * rel(hf_t5_bf, hf_t5) and rel(o_clip_bf[1], hf_clip.pooler_output) <= 1.5 * rel(hf_clip_bf.pooler_output, hf_clip.pooler_output)

    out = {"t5." + k: v.to(torch.bfloat16) for k, v in t5_sd.items()}
    out.update({"clip." + k: v.to(torch.bfloat16) for k, v in clip_sd.items()})
    out.update({"ids_t5": ids_t5, "ids_clip": ids_clip, "hf_t5_fp32": hf_t5, "hf_t5_bf16": hf_t5_bf, "hf_clip_hidden_fp32": hf_clip.last_hidden_state,
                "hf_clip_pooled_fp32": hf_clip.pooler_output, "hf_clip_pooled_bf16": hf_clip_bf.pooler_output.float()})

    # ---- prompt weighting through the UNMODIFIED reference flux_emphasis.py -------------------------------------------------
    pipe = types.SimpleNamespace(name="flux-dev", _execution_device=torch.device("cpu"),
                                 clip=types.SimpleNamespace(tokenizer=clip_tok, hf_module=clip),
                                 t5=types.SimpleNamespace(tokenizer=t5_tok, hf_module=t5))
    meta = {"transformers": transformers.__version__, "prompts": PROMPTS, "parse": {}, "tokens": {}, "groups": {}}
    for s in PARSE_CASES + PROMPTS:
        meta["parse"][s] = ref_emph.parse_prompt_attention(s)
    # seeded random strings over the grammar's alphabet: brackets, escapes, colons, numbers, BREAK, plain words
    import random

    rng = random.Random(1234)
    atoms = ["(", ")", "[", "]", "\\", ":", ":1.5)", ":0.25)", ":-2)", ":+.5)", ":)", " ", "a", "cat", " BREAK ", "BREAK", "1", ".", ",", "\\(", "\\]",
             "sky ", "on", ":x)", "((", "))", "\u00e9"]
    meta["fuzz"] = []
    for _ in range(400):
        text = "".join(rng.choice(atoms) for _ in range(rng.randint(1, 14)))
        try:
            meta["fuzz"].append([text, ref_emph.parse_prompt_attention(text)])
        except ValueError:  # float("1.5.") style payloads: the reference raises
            meta["fuzz"].append([text, "ValueError"])
    for i, prompt in enumerate(PROMPTS):
        tk, tw = ref_emph.get_prompts_tokens_with_weights(clip_tok, prompt)
        t5k, t5w = ref_emph.get_prompts_tokens_with_weights(t5_tok, prompt)
        meta["tokens"][prompt] = {"clip": [tk, tw], "t5": [t5k, t5w]}
        gk, gw = ref_emph.group_tokens_and_weights(list(tk), list(tw), pad_last_block=True, bos=clip_tok.bos_token_id, eos=clip_tok.eos_token_id,
                                                   max_length=77)
        g5k, g5w = ref_emph.group_tokens_and_weights(list(t5k), list(t5w), pad_last_block=True, bos=None, eos=t5_tok.eos_token_id, max_length=512,
                                                     pad_tokens=False)
        meta["groups"][prompt] = {"clip": [gk, gw], "t5": [g5k, g5w]}
        with torch.no_grad():
            vec, txt, txt_ids = ref_emph.get_weighted_text_embeddings_flux(pipe, prompt, num_images_per_prompt=2, device=torch.device("cpu"),
                                                                           target_device=torch.device("cpu"), target_dtype=torch.float32)
        assert vec.shape == (2, 128) and txt.shape == (2, 512, 128) and txt_ids.shape == (2, 512, 3)
        out[f"emph{i}.vec"] = vec[:1].contiguous()
        out[f"emph{i}.txt"] = txt[:1].contiguous()
    # group_tokens_and_weights on a long sequence (chunking)
    long_ids, long_w = list(range(5, 5 + 170)), [1.0 + 0.01 * i for i in range(170)]
    meta["long_group"] = {"in": [long_ids, long_w],
                          "clip": ref_emph.group_tokens_and_weights(list(long_ids), list(long_w), pad_last_block=True, bos=1000, eos=1001, max_length=77),
                          "t5": ref_emph.group_tokens_and_weights(list(long_ids), list(long_w), pad_last_block=True, bos=None, eos=1, max_length=64,
                                                                  pad_tokens=False)}
    # apply_weights / standardize on a random tensor
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(1, 12, 16, generator=g)
    toks = torch.tensor([[5, 6, 7, 8, 1, 0, 0, 0, 0, 0, 0, 0]])
    w = torch.tensor([1.0, 1.5, 1.0, 0.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    out["aw.emb"], out["aw.tokens"], out["aw.weights"] = emb, toks, w
    out["aw.out"] = ref_emph.apply_weights(toks, w, emb.clone(), 1)
    assert torch.equal(to.apply_weights(toks, w, emb, 1), out["aw.out"])
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "g9_text.safetensors"))
    json.dump(meta, open(os.path.join(GOLD, "g9_text.json"), "w"), indent=0)
    print("wrote g9_text.safetensors", os.path.getsize(os.path.join(GOLD, "g9_text.safetensors")), "bytes")


if __name__ == "__main__":
    main()
