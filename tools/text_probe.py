"""Time of one text-conditioning pass at the real model sizes (random weights): T5-v1.1-XXL encoder (24 layers, d_model 4096, 64 heads,
d_ff 10240, L = 512) and CLIP-L text model (12 layers, hidden 768, 12 heads, L = 77).
    python tools/text_probe.py [--iters 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch

from modules.conditioner import ClipTextNative, T5EncoderNative


def randomize(m, dev):
    m.to(dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=dev) * (0.8 / p.shape[-1] ** 0.5)).to(p.dtype))
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t5 = randomize(T5EncoderNative(dict(vocab_size=32128, d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=24,
                                        feed_forward_proj="gated-gelu")), dev)
    clip = randomize(ClipTextNative(dict(vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=12,
                                         max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=49407)), dev)
    ids5 = torch.randint(0, 32128, (1, 512), device=dev)
    idsc = torch.randint(0, 49406, (1, 77), device=dev)
    idsc[0, 30:] = 49407
    for name, m, ids, flops in (("T5-XXL encoder L=512", t5, ids5, 24 * 2 * 512 * (4 * 4096 * 4096 + 3 * 4096 * 10240) + 24 * 64 * 4 * 512 * 512 * 64),
                                ("CLIP-L text L=77", clip, idsc, 12 * 2 * 77 * (4 * 768 * 768 + 2 * 768 * 3072))):
        m(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = m(ids)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        k = "last_hidden_state"
        print(f"{name}: {dt * 1e3:.2f} ms / pass  ({flops / dt / 1e12:.1f} TFLOP/s bf16), out {tuple(out[k].shape)} finite={bool(torch.isfinite(out[k]).all())}",
              flush=True)


if __name__ == "__main__":
    main()
