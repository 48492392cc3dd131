"""stream-K timeline probe: per-workgroup timestamps of one config-17 launch (FLUXMI_SK_ABL=16 + optional ablation bits)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch
from fluxmi import _lib, ops
M, N, K = 4608, 3072, 15360
dev = torch.device("cuda:0")
torch.manual_seed(0)
one = torch.tensor(1.0, device=dev)
a = (torch.randn(M, K, device=dev) * 2).to(torch.float8_e5m2)
w = (torch.randn(N, K, device=dev) * 0.5).to(torch.float8_e4m3fn)
bias = torch.randn(N, device=dev).bfloat16(); gate = torch.randn(N, device=dev).bfloat16()
resid = torch.randn(M, N, device=dev).bfloat16()
run = lambda: ops.linear(a, w, bias, one, one, out=resid, resid=resid, gate=gate, epilogue=_lib.EPI_GATE_RESID, tile_cfg=17)
for _ in range(5): run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 4))()
_lib.lib.fluxmi_gemm_sk_debug.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
_lib.lib.fluxmi_gemm_sk_debug(buf, 256)
raw = torch.tensor(list(buf), dtype=torch.int64).reshape(256, 4)
ticks = (raw[:, 3] >> 8).double()
t = raw.double()
t[:, 3] = (raw[:, 3] & 255).double()
t0 = t[:, 0].min()
for role in (0, 1, 2):
    m = t[:, 3] == role
    if m.sum() == 0: continue
    s, k, e = (t[m, 0] - t0) / 100.0, (t[m, 1] - t0) / 100.0, (t[m, 2] - t0) / 100.0
    clk = ticks[m] / ((t[m, 2] - t[m, 0]) * 10.0)  # shader ticks per ns = GHz
    print(f"role {role}: shader clock GHz mean {clk.mean():.3f} min {clk.min():.3f} max {clk.max():.3f}")
    print(f"role {role}: n={int(m.sum())} start us mean {s.mean():.1f} max {s.max():.1f} | first K loop done mean {k.mean():.1f} min {k.min():.1f} max {k.max():.1f} | end mean {e.mean():.1f} min {e.min():.1f} max {e.max():.1f}")
