"""Prompt weighting ("emphasis") for the Flux text conditioning -- host-side mirror of the reference's flux_emphasis.py.

SURVEY.md §8(f) row 2.  Same public functions, arguments and results as aredden/flux-fp8-api `flux_emphasis.py`
(parse_prompt_attention :12-113, get_prompts_tokens_with_weights :116-172, group_tokens_and_weights :175-243, standardize_tensor
:246-272, apply_weights :275-304, get_weighted_text_embeddings_flux :307-447), written from the grammar / behaviour rather than the
source: a hand-written scanner instead of the reference's regular expression, slicing instead of repeated pops, a masked blend instead
of a per-token loop.  Pinned against the unmodified reference by tests/golden/g9_text.json / g9_text.safetensors
(oracle/gen_golden_text.py).  The encoders it drives are the native ones of modules/conditioner.py.
"""
from __future__ import annotations

import re
from typing import TYPE_CHECKING, List, Optional, Tuple

import torch

if TYPE_CHECKING:
    from flux_pipeline import FluxPipeline

_ROUND_MUL = 1.1        # "(text)"
_SQUARE_MUL = 1 / 1.1   # "[text]"
_BREAK = re.compile(r"\s*\bBREAK\b\s*", re.S)


def _scan(text: str):
    """Token stream of the emphasis grammar, longest alternative first in the reference's order:
    escapes `\\(` `\\)` `\\[` `\\]` `\\\\` (kind "esc", payload = the escaped char) and a lone backslash (payload ""), brackets
    ("open_round" / "open_square" / "close_round" / "close_square"), `:<number>)` ("weight", payload = the number string), a lone ":"
    and runs of any other characters ("text")."""
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if ch == "\\":
            if i + 1 < n and text[i + 1] in "()[]\\":
                yield "esc", text[i + 1]
                i += 2
            else:
                yield "esc", ""
                i += 1
        elif ch == "(":
            yield "open_round", ch
            i += 1
        elif ch == "[":
            yield "open_square", ch
            i += 1
        elif ch == ")":
            yield "close_round", ch
            i += 1
        elif ch == "]":
            yield "close_square", ch
            i += 1
        elif ch == ":":
            j = i + 1
            if j < n and text[j] in "+-":
                j += 1
            k = j
            while k < n and (text[k] == "." or text[k].isdecimal()):
                k += 1
            if k > j and k < n and text[k] == ")":
                yield "weight", text[i + 1:k]
                i = k + 1
            else:
                yield "text", ":"
                i += 1
        else:
            j = i
            while j < n and text[j] not in "\\()[]:":
                j += 1
            yield "text", text[i:j]
            i = j


def parse_prompt_attention(text: str) -> List[list]:
    """'a (red:1.5) cat [dull]' -> [['a ', 1.0], ['red', 1.5], [' cat ', 1.0], ['dull', 0.909...]].

    (abc) multiplies the weight of abc by 1.1, (abc:3.12) by 3.12, [abc] by 1/1.1; brackets nest multiplicatively; unbalanced opening
    brackets apply to the end of the prompt, unbalanced closing brackets are plain text; backslash escapes brackets and itself; the
    word BREAK becomes the marker ['BREAK', -1]; adjacent runs with equal weight are merged."""
    runs: List[list] = []
    open_round: List[int] = []
    open_square: List[int] = []

    def scale_from(start: int, mul: float):
        for r in runs[start:]:
            r[1] *= mul

    for kind, payload in _scan(text):
        if kind == "esc":
            runs.append([payload, 1.0])
        elif kind == "open_round":
            open_round.append(len(runs))
        elif kind == "open_square":
            open_square.append(len(runs))
        elif kind == "weight" and open_round:
            scale_from(open_round.pop(), float(payload))
        elif kind == "close_round" and open_round:
            scale_from(open_round.pop(), _ROUND_MUL)
        elif kind == "close_square" and open_square:
            scale_from(open_square.pop(), _SQUARE_MUL)
        else:
            literal = ":" + payload + ")" if kind == "weight" else payload
            for n, part in enumerate(_BREAK.split(literal)):
                if n:
                    runs.append(["BREAK", -1])
                runs.append([part, 1.0])
    for start in open_round:
        scale_from(start, _ROUND_MUL)
    for start in open_square:
        scale_from(start, _SQUARE_MUL)
    if not runs:
        return [["", 1.0]]
    merged = [runs[0]]
    for r in runs[1:]:
        if r[1] == merged[-1][1]:
            merged[-1][0] += r[0]
        else:
            merged.append(r)
    return merged


def get_prompts_tokens_with_weights(clip_tokenizer, prompt: str, debug: bool = False) -> Tuple[list, list]:
    """Token ids of every weighted run (no special tokens, no truncation) and one weight per token."""
    text_tokens, text_weights = [], []
    for word, weight in parse_prompt_attention(prompt):
        ids = clip_tokenizer(word, truncation=False, padding=False, add_special_tokens=False).input_ids
        if debug:
            print(ids, f"|FOR MODEL LEN{clip_tokenizer.model_max_length}|",
                  clip_tokenizer.decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=True))
        text_tokens.extend(ids)
        text_weights.extend([weight] * len(ids))
    return text_tokens, text_weights


def group_tokens_and_weights(token_ids: list, weights: list, pad_last_block=False, bos=49406, eos=49407, max_length=77, pad_tokens=True):
    """Chunks of `max_len` tokens (max_length - 2 when max_length < 77, else max_length -- the reference's own arithmetic, its TODO
    included), each wrapped in bos/eos with weight 1 when `pad_tokens`; the remainder is wrapped too and, with `pad_last_block`, filled
    with eos up to the chunk size.  Like the reference, the consumed tokens are removed from the caller's lists."""
    max_len = max_length - 2 if max_length < 77 else max_length
    out_ids, out_w = [], []
    n_full = len(token_ids) // max_len
    for c in range(n_full):
        ids, ws = token_ids[c * max_len:(c + 1) * max_len], weights[c * max_len:(c + 1) * max_len]
        if pad_tokens:
            ids, ws = ([bos] if bos is not None else []) + ids + [eos], ([1.0] if bos is not None else []) + ws + [1.0]
        out_ids.append(ids)
        out_w.append(ws)
    del token_ids[:n_full * max_len], weights[:n_full * max_len]
    if token_ids:
        if pad_tokens:
            fill = max_len - len(token_ids) if pad_last_block else 0
            out_ids.append([bos] + token_ids + [eos] * fill + [eos])
            out_w.append([1.0] + weights + [1.0] * fill + [1.0])
        else:
            out_ids.append(token_ids)
            out_w.append(weights)
    return out_ids, out_w


def standardize_tensor(input_tensor: torch.Tensor, target_mean: float, target_std: float) -> torch.Tensor:
    """(x - mean(x)) / std(x) * target_std + target_mean  (global statistics, unbiased std)."""
    return (input_tensor - input_tensor.mean()) / input_tensor.std() * target_std + target_mean


def apply_weights(prompt_tokens: torch.Tensor, weight_tensor: torch.Tensor, token_embedding: torch.Tensor, eos_token_id: int,
                  pad_last_block: bool = True) -> torch.Tensor:
    """Every token whose weight is not 1 is moved along the line through the "pooled" embedding (the first EOS position of each row, or
    the last position) by its weight; the result is re-standardised to the mean / std the embedding had before."""
    mean, std = token_embedding.mean(), token_embedding.std()
    if pad_last_block:
        first_eos = (prompt_tokens.to(dtype=torch.int, device=token_embedding.device) == eos_token_id).int().argmax(dim=-1)
        pooled = token_embedding[torch.arange(token_embedding.shape[0], device=token_embedding.device), first_eos]
    else:
        pooled = token_embedding[:, -1]
    w = weight_tensor.to(token_embedding.device)[: token_embedding.shape[1]]
    pooled = pooled[:, None, :]
    # in the embedding's dtype op by op, as the reference's `pooled + (emb[:, j] - pooled) * weight[j]` with a 0-dim weight tensor does
    blended = pooled + (token_embedding[:, : w.numel()] - pooled) * w.to(token_embedding.dtype)[None, :, None]
    token_embedding = token_embedding.clone()
    token_embedding[:, : w.numel()] = torch.where((w != 1.0)[None, :, None], blended, token_embedding[:, : w.numel()])
    return standardize_tensor(token_embedding, mean, std)


def _flatten(groups):
    return [x for g in groups for x in g]


@torch.inference_mode()
def get_weighted_text_embeddings_flux(pipe: "FluxPipeline", prompt: str = "", num_images_per_prompt: int = 1,
                                      device: Optional[torch.device] = None, target_device: Optional[torch.device] = torch.device("cuda:0"),
                                      target_dtype: Optional[torch.dtype] = torch.bfloat16, debug: bool = False):
    """-> (clip pooled [bs, 768], weighted T5 states [bs, t5_length, 4096], txt_ids zeros [bs, t5_length, 3]).

    The reference's pipeline: weighted token lists from both tokenizers -> grouped / padded -> DECODED back to text and re-tokenised
    with special tokens at the fixed lengths (CLIP 77; T5 512 for flux-dev, 256 otherwise) -> encoders (attention_mask=None) -> the
    per-token weights (T5 only, padded with 1.0) applied by `apply_weights`."""
    device = device or pipe._execution_device
    clip_tok, t5_tok = pipe.clip.tokenizer, pipe.t5.tokenizer
    clip, t5 = pipe.clip.hf_module, pipe.t5.hf_module
    t5_length = 512 if pipe.name == "flux-dev" else 256
    clip_length = 77

    ids_clip, w_clip = get_prompts_tokens_with_weights(clip_tok, prompt, debug=debug)
    ids_t5, w_t5 = get_prompts_tokens_with_weights(t5_tok, prompt, debug=debug)
    g_ids_clip, _ = group_tokens_and_weights(ids_clip, w_clip, pad_last_block=True, bos=clip_tok.bos_token_id, eos=clip_tok.eos_token_id,
                                             max_length=clip_length)
    g_ids_t5, g_w_t5 = group_tokens_and_weights(ids_t5, w_t5, pad_last_block=True, bos=t5_tok.bos_token_id, eos=t5_tok.eos_token_id,
                                                max_length=t5_length, pad_tokens=False)
    text_clip = clip_tok.decode(_flatten(g_ids_clip), skip_special_tokens=True, clean_up_tokenization_spaces=True)
    tokens_clip = clip_tok(text_clip, add_special_tokens=True, padding="max_length", truncation=True, max_length=clip_length,
                           return_tensors="pt").input_ids.to(device)
    text_t5 = t5_tok.decode(_flatten(g_ids_t5), skip_special_tokens=True, clean_up_tokenization_spaces=True)
    tokens_t5 = t5_tok(text_t5, add_special_tokens=True, padding="max_length", truncation=True, max_length=t5_length,
                       return_tensors="pt").input_ids.to(device)
    w = torch.tensor(_flatten(g_w_t5), dtype=torch.float32)
    weights_t5 = torch.cat([w, torch.full((t5_length - w.numel(),), 1.0, dtype=torch.float32)], dim=0).to(device)

    clip_embeds = clip(tokens_clip, output_hidden_states=True, attention_mask=None)["pooler_output"]
    if clip_embeds.shape[0] == 1 and num_images_per_prompt > 1:
        clip_embeds = clip_embeds.expand(num_images_per_prompt, *clip_embeds.shape[1:])
    t5_embeds = t5(tokens_t5, output_hidden_states=True, attention_mask=None)["last_hidden_state"]
    t5_embeds = apply_weights(tokens_t5, weights_t5, t5_embeds, t5_tok.eos_token_id)
    if debug:
        print(t5_embeds.shape)
    if t5_embeds.shape[0] == 1 and num_images_per_prompt > 1:
        t5_embeds = t5_embeds.expand(num_images_per_prompt, *t5_embeds.shape[1:])
    txt_ids = torch.zeros(num_images_per_prompt, t5_embeds.shape[1], 3, device=target_device, dtype=target_dtype)
    return clip_embeds.to(target_device, dtype=target_dtype).contiguous(), t5_embeds.to(target_device, dtype=target_dtype).contiguous(), txt_ids
