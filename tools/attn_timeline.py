"""Which CU ran which attention task / piece when (fluxmi_attention_debug_buffer): the schedule of one launch at the Flux-dev shape, one workgroup per
task against the balanced grid -- round lengths, per-CU busy time, idle gaps, the tail.
    python tools/attn_timeline.py [--L 4608] [--B 1] [--split 0,2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flux-fp8-api_amd"))
import torch

from fluxmi import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=4608)
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--split", default="0,2")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, L = a.B, 24, a.L
Lp = (L + 63) // 64 * 64
q = torch.randn(B, H, L, 128, device=dev).bfloat16()
k = torch.randn(B, H, L, 128, device=dev).half()
vt = torch.randn(B, H, 128, Lp, device=dev).bfloat16()
one = torch.tensor(1.0, device=dev)
o8 = torch.empty(B, L, H * 128, dtype=torch.float8_e5m2, device=dev)
NWG = 4096
for sp in [int(x) for x in a.split.split(",")]:
    with _lib.tuning(attn_split=sp, prefetch=0):
        for _ in range(3):
            ops.attention(q, k, vt, q_scale0=one, out=o8)
        dbg = torch.zeros(NWG * 8, dtype=torch.int64, device=dev)
        _lib.call("fluxmi_attention_debug_buffer", dbg.data_ptr())
        ops.attention(q, k, vt, q_scale0=one, out=o8)
        torch.cuda.synchronize()
        _lib.call("fluxmi_attention_debug_buffer", None)
    t = dbg.cpu().view(NWG, 8)
    t = t[t[:, 3] != 0]
    plan = ops.attention_plan(B, L, H) if sp else None
    if plan and sp == 1 and not plan["thin"]:
        plan = None
    t0 = int(t[:, 2].min())
    start, end = (t[:, 2] - t0).double() * 0.01, (t[:, 3] - t0).double() * 0.01  # us
    cu = ((t[:, 1] & 15) << 16) | (((t[:, 1] >> 8) >> 8) & 0xff)  # XCC id, (SE, SH, CU) bits [15:8] of HW_ID
    print(f"\n== L={L} B={B} attn_split={sp}: {len(t)} workgroups, launch span {float(end.max()):.1f} us; "
          f"{'balanced grid: %d whole + %d pieces per XCD' % (plan['full_per_x'], len(plan['pieces'])) if plan else 'one workgroup per task'}")
    dur = end - start
    ph = [(t[:, i] - t0).double() * 0.01 for i in (4, 5, 6, 7)]
    seg = {"start -> Q fragments built": ph[0] - start, "-> prologue tiles landed": ph[1] - ph[0], "-> step loop done": ph[2] - ph[1], "-> drain done": ph[3] - ph[2],
           "-> end (store / partial + merge)": end - ph[3]}
    print("   phases (median / max us): " + "; ".join(f"{k} {float(v.median()):.2f} / {float(v.max()):.2f}" for k, v in seg.items()))
    print(f"   workgroup duration: min {float(dur.min()):.1f} median {float(dur.median()):.1f} max {float(dur.max()):.1f} us")
    cus = {}
    for i in range(len(t)):
        cus.setdefault(int(cu[i]), []).append((float(start[i]), float(end[i]), int(t[i, 0])))
    busy, lastend, gaps, nwg = [], [], [], []
    for c, lst in cus.items():
        lst.sort()
        busy.append(sum(e - s for s, e, _ in lst))
        lastend.append(lst[-1][1])
        gaps.append(sum(max(0.0, lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)))
        nwg.append(len(lst))
    bt = torch.tensor(busy)
    le = torch.tensor(lastend)
    print(f"   {len(cus)} distinct CUs; workgroups per CU min {min(nwg)} max {max(nwg)}; busy time per CU min {float(bt.min()):.1f} median {float(bt.median()):.1f} max {float(bt.max()):.1f} us; "
          f"last end per CU min {float(le.min()):.1f} median {float(le.median()):.1f} max {float(le.max()):.1f} us; hand-over gaps per CU mean {sum(gaps) / len(gaps):.2f} us")
    if plan:
        n_main = 8 * plan["full_per_x"]
        whole = t[:, 0] < n_main
        if whole.any():
            print(f"   whole tasks: start max {float(start[whole].max()):.1f} us, end min {float(end[whole].min()):.1f} median {float(end[whole].median()):.1f} max {float(end[whole].max()):.1f} us")
        pieces = ~whole
        lens = torch.tensor([plan["pieces"][(int(b) - n_main) >> 3]["len"] for b in t[pieces, 0]], dtype=torch.float64)
        pd = dur[pieces]
        A = torch.stack([lens, torch.ones_like(lens)], 1)
        sol = torch.linalg.lstsq(A, pd.unsqueeze(1)).solution.flatten()
        print(f"   pieces: start min {float(start[pieces].min()):.1f} median {float(start[pieces].median()):.1f} max {float(start[pieces].max()):.1f} us; duration = "
              f"{float(sol[0]):.3f} us per key tile + {float(sol[1]):.1f} us fixed (least squares over {int(pieces.sum())} pieces)")
        # per CU in XCD 0: the sequence
        shown = 0
        for c, lst in sorted(cus.items()):
            if (c >> 16) != 0 or shown >= 6:
                continue
            shown += 1
            seq = ", ".join(f"[{s:.0f}-{e:.0f} {'T' if b < n_main else 'p%d' % plan['pieces'][(b - n_main) >> 3]['len']}]" for s, e, b in lst)
            print(f"     XCD 0 CU {c & 0xffff:#x}: {seq}")
    else:
        whole_d = dur
        nt = (L + 63) // 64
        print(f"   one task = {nt} key tiles: {float(whole_d.median()) / nt:.3f} us per tile incl. prologue / epilogue")
