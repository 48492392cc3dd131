"""CPU restatement of the text-conditioning path (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

SURVEY.md §8(f) row 2.  The reference delegates the two text encoders to a third-party dependency that is NOT under /root/reference:
`transformers` (requirements.txt, unpinned; `T5EncoderModel` for city96/t5-v1_1-xxl-encoder-bf16 and `CLIPTextModel` for
openai/clip-vit-large-patch14, modules/conditioner.py:74-93, called from flux_emphasis.py:420-429 with attention_mask=None).  This
file restates their PUBLISHED algorithms from the HF state-dict layout:

  t5_encoder : T5 v1.1 encoder (Raffel et al. 2020 + "gated-gelu" v1.1 FF): token embedding -> N x [RMS-norm (no mean, no bias) ->
               self-attention WITHOUT 1/sqrt(d) scaling, + bucketed relative-position bias (32 buckets, max distance 128, bidirectional,
               table owned by block 0 and shared by all blocks) -> residual; RMS-norm -> gelu_new(wi_0 x) * wi_1 x -> wo -> residual]
               -> final RMS-norm.  No key-padding mask (the reference passes attention_mask=None): pad tokens attend and are attended.
  clip_text  : CLIP text transformer (Radford et al. 2021): token + learned position embedding -> N x [LayerNorm -> causal self-attention
               (scale 1/sqrt(d), biases) -> residual; LayerNorm -> fc1 -> quick_gelu -> fc2 -> residual] -> final LayerNorm;
               pooled = hidden state at the EOS token (first eos_token_id; legacy eos_token_id == 2 configs: argmax of the ids).

Pinned by oracle/gen_golden_text.py against transformers 5.15.0 (the version in this image) in fp32 on tiny random models, and the
prompt-weighting host logic (flux_emphasis.py) against the UNMODIFIED reference module -> tests/golden/g9_text.safetensors.
`dtype=torch.bfloat16` runs the same graph with bf16 weights / activations (what the reference's `text_enc_dtype: bfloat16` does).
"""
import math

import torch
import torch.nn.functional as F


# ---- T5 ------------------------------------------------------------------------------------------------------------------------
def t5_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """Bidirectional bucketing of (key position - query position): half the buckets per sign; within a sign, exact up to
    num_buckets/4, then log-spaced up to max_distance, clamped to the last bucket."""
    nb = num_buckets // 2
    out = (relative_position > 0).to(torch.long) * nb
    n = relative_position.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


def t5_position_bias(table, L, num_buckets=32, max_distance=128):
    """table [num_buckets, H] (encoder.block.0...relative_attention_bias.weight) -> bias [H, L, L] (query, key)."""
    pos = torch.arange(L, device=table.device)
    bucket = t5_relative_position_bucket(pos[None, :] - pos[:, None], num_buckets, max_distance)
    return table[bucket].permute(2, 0, 1)


def _rms(x, w, eps):
    var = x.float().pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        h = h.to(w.dtype)
    return w * h


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encoder(sd, cfg, input_ids, dtype=torch.float32):
    """cfg: num_layers, num_heads, d_kv, eps (1e-6), num_buckets (32), max_distance (128).  Returns last_hidden_state [B, L, d_model]."""
    g = lambda k: sd[k].to(dtype)
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg.get("eps", 1e-6)
    x = g("encoder.embed_tokens.weight" if "encoder.embed_tokens.weight" in sd else "shared.weight")[input_ids]
    B, L, _ = x.shape
    bias = t5_position_bias(g("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"), L, cfg.get("num_buckets", 32),
                            cfg.get("max_distance", 128))[None]
    for i in range(cfg["num_layers"]):
        p = f"encoder.block.{i}.layer.0."
        h = _rms(x, g(p + "layer_norm.weight"), eps)
        q, k, v = (F.linear(h, g(p + f"SelfAttention.{n}.weight")).view(B, L, H, dk).transpose(1, 2) for n in "qkv")
        s = torch.matmul(q, k.transpose(-1, -2)) + bias  # no 1/sqrt(d): folded into the initialisation of q
        a = F.softmax(s.float(), dim=-1).to(s.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, L, H * dk)
        x = x + F.linear(o, g(p + "SelfAttention.o.weight"))
        p = f"encoder.block.{i}.layer.1."
        h = _rms(x, g(p + "layer_norm.weight"), eps)
        h = gelu_new(F.linear(h, g(p + "DenseReluDense.wi_0.weight"))) * F.linear(h, g(p + "DenseReluDense.wi_1.weight"))
        x = x + F.linear(h, g(p + "DenseReluDense.wo.weight"))
    return _rms(x, g("encoder.final_layer_norm.weight"), eps)


# ---- CLIP text model ---------------------------------------------------------------------------------------------------------------
def clip_text(sd, cfg, input_ids, dtype=torch.float32):
    """cfg: num_layers, num_heads, eps (1e-5), eos_token_id.  Keys as in CLIPTextModel.state_dict(): with the "text_model." prefix
    (checkpoints on the hub, transformers 4.x) or without it (transformers 5.x).  Returns (last_hidden_state [B, L, D],
    pooler_output [B, D])."""
    g = lambda k: (sd["text_model." + k] if ("text_model." + k) in sd else sd[k]).to(dtype)
    H, eps = cfg["num_heads"], cfg.get("eps", 1e-5)
    B, L = input_ids.shape
    x = g("embeddings.token_embedding.weight")[input_ids] + g("embeddings.position_embedding.weight")[:L][None]
    D = x.shape[-1]
    dh = D // H
    causal = torch.full((L, L), float("-inf"), device=input_ids.device).triu(1).to(dtype)
    for i in range(cfg["num_layers"]):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (D,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        q, k, v = (F.linear(h, g(p + f"self_attn.{n}_proj.weight"), g(p + f"self_attn.{n}_proj.bias")).view(B, L, H, dh).transpose(1, 2)
                   for n in "qkv")
        s = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5 + causal
        a = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(o, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (D,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        h = F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))
        h = h * torch.sigmoid(1.702 * h)  # quick_gelu
        x = x + F.linear(h, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    x = F.layer_norm(x, (D,), g("final_layer_norm.weight"), g("final_layer_norm.bias"), eps)
    if cfg.get("eos_token_id", 2) == 2:
        idx = input_ids.to(torch.int).argmax(-1)
    else:
        idx = (input_ids.to(torch.int) == cfg["eos_token_id"]).int().argmax(-1)
    return x, x[torch.arange(B, device=x.device), idx]


# ---- prompt weighting (flux_emphasis.py:267-304) -----------------------------------------------------------------------------------
def apply_weights(prompt_tokens, weight_tensor, token_embedding, eos_token_id, pad_last_block=True):
    """flux_emphasis.py:281-304: every token whose weight != 1 is moved away from / towards the EOS ("pooled") embedding by its weight,
    then the whole tensor is re-standardised to its original mean / std (:267-278, unbiased std)."""
    token_embedding = token_embedding.clone()
    mean, std = token_embedding.mean(), token_embedding.std()
    if pad_last_block:
        idx = (prompt_tokens.to(torch.int) == eos_token_id).int().argmax(-1)
        pooled = token_embedding[torch.arange(token_embedding.shape[0]), idx]
    else:
        pooled = token_embedding[:, -1]
    for j in range(len(weight_tensor)):
        if weight_tensor[j] != 1.0:
            token_embedding[:, j] = pooled + (token_embedding[:, j] - pooled) * weight_tensor[j]
    m2, s2 = token_embedding.mean(), token_embedding.std()
    return (token_embedding - m2) / s2 * std + mean
