"""VAE decode at the real FLUX geometry (ch 128, [1,2,4,4], z 16) on a 1024x1024 image: wall time per image and peak memory.
    python tools/probes/vae_probe.py [--size 1024] [--iters 5]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "flux-fp8-api_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch

import vae_oracle as vo
from modules.autoencoder import AutoEncoder, AutoEncoderParams

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
ae = AutoEncoder(AutoEncoderParams(**vo.FULL_PARAMS))
sd = vo.synth_state_dict({k: v.shape for k, v in ae.state_dict().items()}, seed=7)
ae.load_state_dict(sd, strict=True)
ae.to(dev)
z = torch.randn(1, 16, a.size // 8, a.size // 8, device=dev)
with torch.inference_mode():
    out = ae.decode(z)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = ae.decode(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
print(f"VAE decode {a.size}x{a.size}: {dt * 1e3:.2f} ms per image, peak allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, output {tuple(out.shape)} "
      f"finite {bool(torch.isfinite(out.float()).all())} checksum {out.float().abs().mean().item():.6f}")
