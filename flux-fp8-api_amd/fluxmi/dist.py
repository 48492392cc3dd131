"""Batch-sharded replicas over RCCL/xGMI: one process per GPU, a full fp8 weight replica on each,
sample i of a request on rank floor(i*world/B)... exactly one exchange before the loop and one after it:

  broadcast_request : the conditioning produced on the text-encoder rank -- T5 states `txt` [B,Lt,4096],
                      CLIP pooled `vec` [B,768] -- plus the packed request noise [B,Li,64] so that a batch of 8 on
                      8 GPUs reproduces the same batch on 1 GPU; ONE flat broadcast (payload <= 34 MB: latency-,
                      not bandwidth-bound, so no ring/bucketing)
  gather_latents    : final latents [B_local,Li,64] back to the VAE rank

There is no per-step collective: batch elements never interact inside Flux.forward.  During the 12 calibration
steps after load, `allreduce_amax` keeps the F8Linear input scales identical on every replica (the reference
computes amax over the whole batch, float8_quantize.py:227).  backend "nccl" IS RCCL on ROCm; the CPU tests run
the same code over "gloo".
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as td


def is_dist() -> bool:
    return td.is_available() and td.is_initialized()


def world_size() -> int:
    return td.get_world_size() if is_dist() else 1


def rank() -> int:
    return td.get_rank() if is_dist() else 0


def shard_bounds(batch: int, rank_: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `batch` samples over `world` ranks (ranks beyond `batch` get an empty slice)."""
    base, rem = divmod(batch, world)
    lo = rank_ * base + min(rank_, rem)
    return lo, lo + base + (1 if rank_ < rem else 0)


def broadcast_request(txt: torch.Tensor, vec: torch.Tensor, noise: torch.Tensor, src: int = 0):
    """One flat buffer = [txt | vec | noise] (same dtype) broadcast from `src`; shapes must already agree on all ranks."""
    if not is_dist():
        return txt, vec, noise
    parts = [txt, vec, noise]
    flat = torch.cat([p.reshape(-1) for p in parts])
    td.broadcast(flat, src=src)
    out, off = [], 0
    for p in parts:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return tuple(out)


def gather_latents(lat: torch.Tensor, batch: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate per-rank latent shards on `dst` (None elsewhere)."""
    if not is_dist():
        return lat
    world, r = world_size(), rank()
    shapes = [shard_bounds(batch, i, world) for i in range(world)]
    bufs = [torch.empty((hi - lo,) + tuple(lat.shape[1:]), dtype=lat.dtype, device=lat.device) for lo, hi in shapes]
    td.all_gather(bufs, lat) if all(b.shape == bufs[0].shape for b in bufs) else _uneven_all_gather(bufs, lat)
    return torch.cat(bufs, 0) if r == dst else None


def _uneven_all_gather(bufs, lat):
    for i, b in enumerate(bufs):
        if i == rank():
            b.copy_(lat)
        if b.numel():
            td.broadcast(b, src=i)


def allreduce_amax(amax: torch.Tensor) -> torch.Tensor:
    """MAX all-reduce of a vector of per-layer activation amax values (calibration steps only)."""
    if is_dist():
        td.all_reduce(amax, op=td.ReduceOp.MAX)
    return amax


def sync_calibration(f8_modules, model=None) -> int:
    """(Fallback; the exact scheme is Flux.enable_amax_exchange, which reduces every layer's amax INSIDE each calibrating step so
    that later layers also see identically scaled inputs.)  Make the frozen F8Linear input scales identical on every replica: MAX all-reduce of every layer's 12 running-amax trials
    (one [n_layers, n_trials] tensor, ~15 KB), then input_scale = amax_to_scale(max(trials)) exactly as the reference computes it
    (float8_quantize.py:214-215,237-246: python_float / tensor, clamped at the format maximum).  The reference takes amax over the
    whole batch (float8_quantize.py:227); with one sample per GPU this is what makes a batch of N on N GPUs calibrate like a batch of
    N on one GPU.  Call after the calibration steps; a no-op (returns 0) outside a process group.  Returns the number of layers."""
    mods = [m for m in f8_modules if getattr(m, "input_amax_trials", None) is not None and m.input_scale is not None]
    if not mods or not is_dist():
        return 0
    trials = torch.stack([m.input_amax_trials.float() for m in mods])
    td.all_reduce(trials, op=td.ReduceOp.MAX)
    for m, t in zip(mods, trials):
        m.input_amax_trials.copy_(t)
        max_val = float(m.input_max_value)
        scale = (max_val / torch.clamp(t.max(), min=1e-12)).clamp(max=max_val)
        m.input_scale.copy_(scale)
        m.input_scale_reciprocal.copy_(scale.reciprocal())
    # the engine caches state derived from the scales (64 KiB quantising-epilogue tables, the captured hipGraph that reads them):
    # pass `model` (a modules.flux_model.Flux) to have it rebuilt; callers that do not MUST call model.rebind_weights() themselves
    if model is not None and hasattr(model, "rebind_weights"):
        model.rebind_weights()
    return len(mods)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """torchrun-style bootstrap: returns (rank, world, local_rank)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank_ = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        td.init_process_group(backend=backend, rank=rank_, world_size=world)
    return rank_, world, local
