"""Which of the ORACLE's host ops give other bits on another host CPU?  (TEST INFRASTRUCTURE; the oracle is torch on the host cores.)
The build container is an Intel Xeon with AMX, the GPU box an AMD EPYC with AVX-512 bf16: torch / oneDNN / MKL pick other kernels, and the
full-geometry fixtures (pinned to the reference in the build container) are only reproduced on 8 of 92 tensors on the GPU box.
Prints a hash per op family; run it on both hosts, with and without ONEDNN_MAX_CPU_ISA=AVX512_CORE_BF16, and compare.
    python tools/probes/host_ops_probe.py"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flux-fp8-api_amd"), os.path.join(ROOT, "oracle")]
import torch
import torch.nn.functional as F

import flux_oracle as fo

h = lambda t: hashlib.sha1(t.detach().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:10]
torch.manual_seed(0)
out = {}
x = torch.randn(1024, 3072).bfloat16()
w = (torch.randn(3072, 3072) * 0.02).bfloat16()
b = torch.randn(3072).bfloat16()
out["bf16 linear 1024x3072x3072"] = h(F.linear(x, w, b))
out["bf16 linear M=1"] = h(F.linear(x[:1], w, b))
wl = (torch.randn(3072, 15360) * 0.02).bfloat16()
xl = torch.randn(512, 15360).bfloat16()
out["bf16 linear K=15360"] = h(F.linear(xl, wl, b))
q, k, v = (torch.randn(1, 4, 1024, 128).bfloat16() for _ in range(3))
out["sdpa bf16 L=1024"] = h(F.scaled_dot_product_attention(q, k, v))
q2, k2, v2 = (torch.randn(1, 1, 4608, 128).bfloat16() for _ in range(3))
out["sdpa bf16 L=4608"] = h(F.scaled_dot_product_attention(q2, k2, v2))
st = fo.F8LinearState(w, b)
out["F8Linear oracle call"] = h(st(x))
x8 = (x.float() * 3).clamp(-57344, 57344).to(torch.float8_e5m2)
w8 = (w.float() * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
one = torch.tensor(1.0)
out["_scaled_mm e5m2 x e4m3"] = h(torch._scaled_mm(x8, w8.T, scale_a=one, scale_b=one, bias=b, out_dtype=torch.bfloat16))
out["fp32 matmul 1024x3072x3072"] = h(x.float() @ w.float().T)
out["layer_norm"] = h(F.layer_norm(x, (3072,), eps=1e-6))
out["rms_norm fp32"] = h(F.rms_norm(q.float(), (128,), torch.ones(128), eps=1e-6).to(q))
out["gelu tanh"] = h(F.gelu(x, approximate="tanh"))
out["silu"] = h(F.silu(x))
out["amax"] = h(x.abs().max().float().reshape(1))
t = torch.linspace(0, 1, 29).bfloat16()
out["timestep embedding"] = h(fo.timestep_embedding(t, 256))
ids = torch.zeros(1, 64, 3).bfloat16()
ids[..., 1] = torch.arange(64).bfloat16()
out["rope table"] = h(fo.rope_table(ids, [16, 56, 56], 10000, torch.bfloat16))
out["exp / cos fp32"] = h(torch.cat((torch.exp(x.float().flatten()[:4096]), torch.cos(x.float().flatten()[:4096]))))
cpu = os.popen("lscpu | grep 'Model name'").read().strip().split(":")[-1].strip()
print(f"# host: {cpu}; torch {torch.__version__}; threads {torch.get_num_threads()}; ONEDNN_MAX_CPU_ISA={os.environ.get('ONEDNN_MAX_CPU_ISA')}")
for kx, vx in out.items():
    print(f"{kx:32s} {vx}")
