"""Turn a rocprofv3 (ROCm 7.2) kernel trace -- rocpd sqlite output (*.db) or `--output-format csv` (*kernel_trace.csv) -- into the
plain-text per-kernel summary that is committed under profiles/:
    python tools/rocprof_summary.py <dir-with-*.db or *_kernel_trace.csv> [-o profiles/xyz.txt] [--steady]"""
import argparse
import csv
import glob
import os
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if name.startswith("at::native") or "at::native" in name[:40]:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", name)
        return "torch::" + (m.group(1) if m else "kernel") + "<...>"
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--header", default="")
    ap.add_argument("--steady", action="store_true",
                    help="only the fused (frozen-scale) denoise steps: dispatches after the last calibration kernel up to the last "
                         "euler_kernel; prints per-step time and launch count per kernel")
    args = ap.parse_args()
    if args.steady and (os.path.isdir(args.path) and glob.glob(os.path.join(args.path, "**", "*kernel_trace.csv"), recursive=True) or args.path.endswith(".csv")):
        # CSV traces: the one implementation bench.py uses for roofline.step (steady steps of the LAST request, per-family totals)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench

        s = bench.summarize_step_trace(args.path, header=args.header)
        if s is None:
            sys.exit("no steady denoise steps found in " + args.path)
        if args.out:
            os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
            open(args.out, "w").write(s["text"])
        print(s["text"])
        return
    dbs = glob.glob(os.path.join(args.path, "**", "*.db"), recursive=True) if os.path.isdir(args.path) else [args.path]
    if not dbs and os.path.isdir(args.path):
        dbs = glob.glob(os.path.join(args.path, "**", "*kernel_trace.csv"), recursive=True)
    if not dbs:
        sys.exit("no .db / *kernel_trace.csv found under " + args.path)
    rows = {}
    steps = 0
    span_ns = 0
    for db in dbs:
        if db.endswith(".csv"):
            ks = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(db))), key=lambda k: k[1])
            if not args.steady:
                for n, s_, e in ks:
                    k = short(n)
                    c, t = rows.get(k, (0, 0.0))
                    rows[k] = (c + 1, t + (e - s_))
                continue
        else:
            cur = sqlite3.connect(db).cursor()
        if args.steady:
            if not db.endswith(".csv"):
                ks = list(cur.execute("select name, start, end from kernels order by start"))
            calib_end = max([e for n, s_, e in ks if "amax_kernel" in n or "calib_update" in n] or [0])
            eul = [e for n, s_, e in ks if "euler_kernel" in n and s_ > calib_end]
            if not eul:
                continue
            # drop the first fused step (runs eagerly before the graph capture)
            t0 = eul[0] if len(eul) > 1 else calib_end
            t1 = eul[-1]
            steps += len(eul) - 1 if len(eul) > 1 else 1
            span_ns += t1 - t0
            for n, s_, e in ks:
                if s_ >= t0 and e <= t1:
                    k = short(n)
                    c, t = rows.get(k, (0, 0.0))
                    rows[k] = (c + 1, t + (e - s_))
            continue
        for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            k = short(name)
            c, t = rows.get(k, (0, 0.0))
            rows[k] = (c + calls, t + total)
    tot = sum(t for _, t in rows.values())
    lines = [args.header] if args.header else []
    if args.steady and steps:
        lines.append(f"# steady state: {steps} fused graph-replayed denoise steps, wall {span_ns / steps / 1e6:.3f} ms/step, "
                     f"kernel time {tot / steps / 1e6:.3f} ms/step, {sum(c for c, _ in rows.values()) / steps:.0f} launches/step; "
                     "columns are totals over those steps (ns resolution -> us)")
        rows = {k: (c, t / 1e3) for k, (c, t) in rows.items()}
        tot /= 1e3
    lines.append(f"{'kernel':112s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:112s} {c:7d} {t:12.1f} {t / c:10.2f} {100 * t / tot:6.2f}")
    text = "\n".join(lines) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        open(args.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
