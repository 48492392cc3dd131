"""Text conditioning on MI355X -- `HFEmbedder` of the reference (modules/conditioner.py of aredden/flux-fp8-api) over native encoders.

SURVEY.md §8(f) row 2.  The reference wraps transformers' `T5EncoderModel` (city96/t5-v1_1-xxl-encoder-bf16) and `CLIPTextModel`
(openai/clip-vit-large-patch14) plus their tokenizers (conditioner.py:74-93) and calls them with `attention_mask=None`
(conditioner.py:102-117, flux_emphasis.py:420-429).  Here `hf_module` is a native module with the SAME state-dict keys and the same
call convention (`hf_module(input_ids, attention_mask=None, output_hidden_states=...)["last_hidden_state" | "pooler_output"]`):

  * every Linear runs on the bf16 MFMA GEMM of libfluxmi (q|k and wi_0|wi_1 fused into one GEMM each, residual adds in the GEMM's
    gate*y+x epilogue, V produced already transposed by swapping the v-projection's operands);
  * T5LayerNorm / LayerNorm = `fluxmi_row_norm`, gelu_new-gate / quick_gelu = `fluxmi_act_mul`, attention = `fluxmi_text_attention`
    (head_dim 64; T5: no scaling + bucketed relative-position bias as a function of key - query; CLIP: 1/sqrt(d) + causal mask);
  * embedding lookups and the EOS gather are torch indexing (plumbing).

Tokenizers stay with `transformers` (pure host code).  Weights are kept in bf16 (`text_enc_dtype`); the reference's
`quantization_dtype` knobs (quanto qfloat8 / bitsandbytes qint4 ...: weight-only storage formats whose matmuls still run in bf16) are
accepted and ignored -- 9.4 GB of bf16 T5 weights is 3 % of this GPU's HBM.  There is no CPU / PyTorch fallback.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import torch
from torch import Tensor, nn


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


class _Weight(nn.Module):
    """A module that only owns `weight` (T5LayerNorm / relative_attention_bias / embeddings keep the HF key names)."""

    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(*shape), requires_grad=False)


def _lin(i, o, bias):
    m = nn.Linear(i, o, bias=bias)
    m.requires_grad_(False)
    return m


class _Cache:
    """Fused / derived weights, rebuilt when the source parameter object or device changes."""

    def __init__(self):
        self._d = {}

    def get(self, key, srcs, fn):
        ent = self._d.get(key)
        if ent is None or any(a is not b for a, b in zip(ent[0], srcs)) or ent[1].device != srcs[0].device:
            ent = (tuple(srcs), fn())
            self._d[key] = ent
        return ent[1]


def _bf(t: Tensor) -> Tensor:
    return t.detach().to(torch.bfloat16).contiguous()


def _gemm(a, w, bias=None, resid=None, ones=None):
    from fluxmi import _lib, ops

    if resid is None:
        return ops.linear(a, w, bias)
    return ops.linear(a, w, bias, epilogue=_lib.EPI_GATE_RESID, gate=ones, resid=resid, out=torch.empty_like(resid))  # resid + (a w^T + b)


# ---- T5 v1.1 encoder -----------------------------------------------------------------------------------------------------------------
class T5EncoderNative(nn.Module):
    """State-dict keys of transformers' T5EncoderModel: shared / encoder.embed_tokens, encoder.block.{i}.layer.0.SelfAttention.{q,k,v,o},
    encoder.block.0.layer.0.SelfAttention.relative_attention_bias, encoder.block.{i}.layer.{0,1}.layer_norm,
    encoder.block.{i}.layer.1.DenseReluDense.{wi_0,wi_1,wo}, encoder.final_layer_norm."""

    def __init__(self, config: dict):
        super().__init__()
        c = self.cfg = dict(config)
        if c.get("feed_forward_proj", "gated-gelu") != "gated-gelu" or c.get("d_kv", 64) != 64:
            raise ValueError("fluxmi: the native T5 encoder covers T5 v1.1 (gated-gelu FF) with d_kv = 64")
        D, H, Fd, V = c["d_model"], c["num_heads"], c["d_ff"], c["vocab_size"]
        self.shared = _Weight(V, D)
        self.encoder = nn.Module()
        self.encoder.embed_tokens = _Weight(V, D)
        self.encoder.block = nn.ModuleList()
        for i in range(c["num_layers"]):
            blk = nn.Module()
            l0, l1 = nn.Module(), nn.Module()
            l0.SelfAttention = nn.Module()
            for n in "qkv":
                setattr(l0.SelfAttention, n, _lin(D, H * 64, False))
            l0.SelfAttention.o = _lin(H * 64, D, False)
            if i == 0:
                l0.SelfAttention.relative_attention_bias = _Weight(c.get("relative_attention_num_buckets", 32), H)
            l0.layer_norm = _Weight(D)
            l1.DenseReluDense = nn.Module()
            l1.DenseReluDense.wi_0, l1.DenseReluDense.wi_1, l1.DenseReluDense.wo = _lin(D, Fd, False), _lin(D, Fd, False), _lin(Fd, D, False)
            l1.layer_norm = _Weight(D)
            blk.layer = nn.ModuleList([l0, l1])
            self.encoder.block.append(blk)
        self.encoder.final_layer_norm = _Weight(D)
        self._cache = _Cache()

    def load_state_dict(self, sd, strict=True, assign=False):
        sd = dict(sd)
        if "encoder.embed_tokens.weight" not in sd and "shared.weight" in sd:  # tied in the checkpoint
            sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd["shared.weight"] = sd["encoder.embed_tokens.weight"]
        return super().load_state_dict(sd, strict=strict, assign=assign)

    @property
    def device(self):
        return self.encoder.final_layer_norm.weight.device

    def _rel_bias(self, Lp: int) -> Tensor:
        """fp32 [H, 2*Lp]: column key - query + Lp -> relative_attention_bias[bucket(key - query)] (bidirectional bucketing, Raffel et al.
        2020: half the buckets per sign; exact below num_buckets/4, log-spaced up to max_distance beyond)."""
        tab = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight

        def build():
            nb, maxd = self.cfg.get("relative_attention_num_buckets", 32) // 2, self.cfg.get("relative_attention_max_distance", 128)
            d = torch.arange(-Lp, Lp)
            n = d.abs()
            exact = nb // 2
            large = exact + (torch.log(n.float() / exact) / math.log(maxd / exact) * (nb - exact)).to(torch.long)
            bucket = (d > 0).long() * nb + torch.where(n < exact, n, torch.minimum(large, torch.full_like(large, nb - 1)))
            return tab.detach().to(torch.bfloat16).float()[bucket.to(tab.device)].t().contiguous()

        return self._cache.get(("rel", Lp), [tab], build)

    @torch.inference_mode()
    def forward(self, input_ids: Tensor, attention_mask=None, output_hidden_states: bool = False, **_):
        from fluxmi import ops

        if attention_mask is not None:
            raise NotImplementedError("fluxmi: the reference calls the text encoders with attention_mask=None (conditioner.py:112-116)")
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("fluxmi: the text encoders need the GPU (libfluxmi has no CPU path)")
        c, ck = self.cfg, self._cache
        H, eps = c["num_heads"], c.get("layer_norm_epsilon", 1e-6)
        input_ids = input_ids.to(dev).view(-1, input_ids.shape[-1])
        B, L = input_ids.shape
        Lp = _pad32(L)
        emb = ck.get("emb", [self.encoder.embed_tokens.weight], lambda: _bf(self.encoder.embed_tokens.weight))
        ones = ck.get("ones", [self.encoder.final_layer_norm.weight], lambda: torch.ones(c["d_model"], dtype=torch.bfloat16, device=dev))
        rel = self._rel_bias(Lp)
        outs = []
        for b in range(B):
            x = torch.zeros(Lp, c["d_model"], dtype=torch.bfloat16, device=dev)
            x[:L] = emb[input_ids[b]]
            for i, blk in enumerate(self.encoder.block):
                at, ff = blk.layer[0], blk.layer[1].DenseReluDense
                sa = at.SelfAttention
                wqk = ck.get(("qk", i), [sa.q.weight, sa.k.weight], lambda: torch.cat([_bf(sa.q.weight), _bf(sa.k.weight)], 0))
                wv = ck.get(("v", i), [sa.v.weight], lambda: _bf(sa.v.weight))
                wo = ck.get(("o", i), [sa.o.weight], lambda: _bf(sa.o.weight))
                wi = ck.get(("wi", i), [ff.wi_0.weight, ff.wi_1.weight], lambda: torch.cat([_bf(ff.wi_0.weight), _bf(ff.wi_1.weight)], 0))
                wwo = ck.get(("wo", i), [ff.wo.weight], lambda: _bf(ff.wo.weight))
                ln0 = ck.get(("ln0", i), [at.layer_norm.weight], lambda: _bf(at.layer_norm.weight))
                ln1 = ck.get(("ln1", i), [blk.layer[1].layer_norm.weight], lambda: _bf(blk.layer[1].layer_norm.weight))
                h = ops.row_norm(x, ln0, eps=eps, rms=True)
                qk = _gemm(h, wqk)                                    # [Lp, 2*H*64]
                vt = _gemm(wv, h)                                     # [H*64, Lp] = W_v . h^T = V^T
                o = ops.text_attention(qk[:, : H * 64], qk[:, H * 64:], vt, L, H, scale=1.0, rel_bias=rel)
                x = _gemm(o, wo, resid=x, ones=ones)
                h = ops.row_norm(x, ln1, eps=eps, rms=True)
                x = _gemm(ops.act_mul(_gemm(h, wi), gated=True), wwo, resid=x, ones=ones)
            fin = ck.get("fin", [self.encoder.final_layer_norm.weight], lambda: _bf(self.encoder.final_layer_norm.weight))
            outs.append(ops.row_norm(x, fin, eps=eps, rms=True)[:L])
        return {"last_hidden_state": torch.stack(outs)}


# ---- CLIP text model -----------------------------------------------------------------------------------------------------------------
class _ClipTextTransformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        D, H, Fd = c["hidden_size"], c["num_attention_heads"], c["intermediate_size"]
        self.embeddings = nn.Module()
        self.embeddings.token_embedding = _Weight(c["vocab_size"], D)
        self.embeddings.position_embedding = _Weight(c.get("max_position_embeddings", 77), D)
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList()
        for _ in range(c["num_hidden_layers"]):
            lay = nn.Module()
            lay.self_attn = nn.Module()
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                setattr(lay.self_attn, n, _lin(D, D, True))
            lay.layer_norm1, lay.layer_norm2 = nn.LayerNorm(D), nn.LayerNorm(D)
            lay.mlp = nn.Module()
            lay.mlp.fc1, lay.mlp.fc2 = _lin(D, Fd, True), _lin(Fd, D, True)
            self.encoder.layers.append(lay)
        self.final_layer_norm = nn.LayerNorm(D)
        self.requires_grad_(False)


class ClipTextNative(nn.Module):
    """State-dict keys of transformers' CLIPTextModel: text_model.embeddings.{token,position}_embedding,
    text_model.encoder.layers.{i}.{self_attn.{q,k,v,out}_proj, layer_norm1, layer_norm2, mlp.fc1, mlp.fc2}, text_model.final_layer_norm
    (keys without the `text_model.` prefix -- the transformers 5.x layout -- are accepted too)."""

    def __init__(self, config: dict):
        super().__init__()
        # config.json of a CLIP checkpoint nests the text tower (`text_config`, older files also `text_config_dict`); keys left out mean
        # transformers' CLIPTextConfig defaults
        c = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=8,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2)
        if "text_config" in config or "text_config_dict" in config:
            c.update(config.get("text_config_dict") or {})
            c.update(config.get("text_config") or {})
        else:
            c.update(config)
        self.cfg = c
        if c.get("hidden_act", "quick_gelu") != "quick_gelu" or c["hidden_size"] // c["num_attention_heads"] != 64:
            raise ValueError("fluxmi: the native CLIP text model covers quick_gelu MLPs with head_dim = 64 (CLIP ViT-L/14, ViT-B)")
        self.text_model = _ClipTextTransformer(c)
        self._cache = _Cache()

    def load_state_dict(self, sd, strict=True, assign=False):
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in sd.items()
              if not k.endswith("position_ids") and not k.startswith(("vision_model.", "visual_projection", "text_projection", "logit_scale"))}
        return super().load_state_dict(sd, strict=strict, assign=assign)

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    @torch.inference_mode()
    def forward(self, input_ids: Tensor, attention_mask=None, output_hidden_states: bool = False, **_):
        from fluxmi import ops

        if attention_mask is not None:
            raise NotImplementedError("fluxmi: the reference calls the text encoders with attention_mask=None (conditioner.py:112-116)")
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("fluxmi: the text encoders need the GPU (libfluxmi has no CPU path)")
        c, ck, tm = self.cfg, self._cache, self.text_model
        D, H, eps = c["hidden_size"], c["num_attention_heads"], c.get("layer_norm_eps", 1e-5)
        input_ids = input_ids.to(dev).view(-1, input_ids.shape[-1])
        B, L = input_ids.shape
        Lp = _pad32(L)
        tok = ck.get("tok", [tm.embeddings.token_embedding.weight], lambda: _bf(tm.embeddings.token_embedding.weight))
        pos = ck.get("pos", [tm.embeddings.position_embedding.weight], lambda: _bf(tm.embeddings.position_embedding.weight))
        ones = ck.get("ones", [tm.final_layer_norm.weight], lambda: torch.ones(D, dtype=torch.bfloat16, device=dev))
        hid, pooled = [], []
        for b in range(B):
            x = torch.zeros(Lp, D, dtype=torch.bfloat16, device=dev)
            x[:L] = tok[input_ids[b]] + pos[:L]
            for i, lay in enumerate(tm.encoder.layers):
                sa, mlp = lay.self_attn, lay.mlp
                wqk = ck.get(("qk", i), [sa.q_proj.weight, sa.k_proj.weight], lambda: torch.cat([_bf(sa.q_proj.weight), _bf(sa.k_proj.weight)], 0))
                bqk = ck.get(("bqk", i), [sa.q_proj.bias, sa.k_proj.bias], lambda: torch.cat([_bf(sa.q_proj.bias), _bf(sa.k_proj.bias)], 0))
                p = {}
                for name, t in (("wv", sa.v_proj.weight), ("bv", sa.v_proj.bias), ("wo", sa.out_proj.weight), ("bo", sa.out_proj.bias),
                                ("w1", mlp.fc1.weight), ("b1", mlp.fc1.bias), ("w2", mlp.fc2.weight), ("b2", mlp.fc2.bias),
                                ("g1", lay.layer_norm1.weight), ("e1", lay.layer_norm1.bias), ("g2", lay.layer_norm2.weight),
                                ("e2", lay.layer_norm2.bias)):
                    p[name] = ck.get((name, i), [t], lambda t=t: _bf(t))
                h = ops.row_norm(x, p["g1"], p["e1"], eps=eps, rms=False)
                qk = _gemm(h, wqk, bqk)
                vt = _gemm(p["wv"], h)                                # V^T without its bias: rows of P sum to 1 -> added after P V
                o = ops.text_attention(qk[:, :D], qk[:, D:], vt, L, H, scale=64 ** -0.5, causal=True, v_bias=p["bv"])
                x = _gemm(o, p["wo"], p["bo"], resid=x, ones=ones)
                h = ops.row_norm(x, p["g2"], p["e2"], eps=eps, rms=False)
                x = _gemm(ops.act_mul(_gemm(h, p["w1"], p["b1"]), gated=False), p["w2"], p["b2"], resid=x, ones=ones)
            g, e = (ck.get((n, "fin"), [t], lambda t=t: _bf(t)) for n, t in (("g", tm.final_layer_norm.weight), ("e", tm.final_layer_norm.bias)))
            x = ops.row_norm(x, g, e, eps=eps, rms=False)[:L]
            ids = input_ids[b].to(torch.int)
            eos = c.get("eos_token_id", 2)
            idx = ids.argmax(-1) if eos == 2 else (ids == eos).int().argmax(-1)  # legacy configs (eos_token_id == 2): highest id = EOT
            hid.append(x)
            pooled.append(x[idx])
        return {"last_hidden_state": torch.stack(hid), "pooler_output": torch.stack(pooled)}


# ---- the reference's wrapper ---------------------------------------------------------------------------------------------------------
def _read_dir_weights(path: str) -> dict:
    from safetensors.torch import load_file

    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"fluxmi: no *.safetensors weights under {path}")
    sd = {}
    for f in files:
        sd.update(load_file(os.path.join(path, f), device="cpu"))
    return sd


class HFEmbedder(nn.Module):
    """reference modules/conditioner.py:38-117, same constructor arguments.  `version` is a local directory in the HF layout
    (config.json + *.safetensors + tokenizer files; there is no network here).  Offline / test hooks: `hf_config=` (dict),
    `state_dict=`, `tokenizer=` replace what would be read from `version`."""

    def __init__(self, version: str, max_length: int, device, quantization_dtype: Optional[str] = None, offloading_device=torch.device("cpu"),
                 is_clip: bool = False, hf_config: Optional[dict] = None, state_dict: Optional[dict] = None, tokenizer=None, **hf_kwargs):
        super().__init__()
        self.offloading_device = offloading_device if isinstance(offloading_device, torch.device) else torch.device(offloading_device)
        self.device = device if isinstance(device, torch.device) else torch.device(device)
        self.is_clip = str(version).startswith("openai") or is_clip
        self.max_length = max_length
        self.output_key = "pooler_output" if self.is_clip else "last_hidden_state"
        self.quantization_dtype = quantization_dtype  # accepted, ignored: weights stay bf16 (see the module docstring)
        if tokenizer is None:
            from transformers import CLIPTokenizer, T5Tokenizer

            tokenizer = (CLIPTokenizer if self.is_clip else T5Tokenizer).from_pretrained(version, max_length=max_length)
        self.tokenizer = tokenizer
        if hf_config is None:
            with open(os.path.join(version, "config.json")) as f:
                hf_config = json.load(f)
        self.hf_module = (ClipTextNative if self.is_clip else T5EncoderNative)(hf_config)
        if state_dict is None:
            state_dict = _read_dir_weights(version)
        missing, unexpected = self.hf_module.load_state_dict(state_dict, strict=False)
        if missing:
            raise RuntimeError(f"fluxmi: text-encoder checkpoint is missing weights: {missing[:4]} ...")
        self.hf_module.to(device=self.device, dtype=torch.bfloat16)

    def offload(self):  # reference :95-97 -- meaningless with 288 GB of HBM; kept as a no-op
        pass

    def cuda(self):  # reference :99-100
        self.hf_module.to(device=self.device)

    def forward(self, text: list[str]) -> Tensor:
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=False, return_overflowing_tokens=False,
                             padding="max_length", return_tensors="pt")
        out = self.hf_module(input_ids=enc["input_ids"].to(self.hf_module.device), attention_mask=None, output_hidden_states=False)
        return out[self.output_key]
