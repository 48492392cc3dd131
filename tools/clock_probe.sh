#!/bin/bash
# effective clock + cycles of GEMM variants: tools/clock_probe.sh <outdir> <shape> <cfg...>
OUT=$1; SHAPE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
for c in "$@"; do
  (cd $R && timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/$OUT/c$c -- python tools/gemm_probe.py --shape $SHAPE --cfg $c --iters 8) > $R/$OUT/c$c.log 2>&1
done
cd $R
python - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in sys.argv[2:]:
    kt = glob.glob(f"{out}/c{c}/*/*kernel_trace.csv"); cc = glob.glob(f"{out}/c{c}/*/*counter_collection.csv")
    if not kt or not cc: print(c, "no data"); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if "gemm" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ctr = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        if "gemm" in r["Kernel_Name"]: ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(set(dur) & set(ctr), key=int)[2:]
    if not ids: print(c, "no dispatch"); continue
    n = len(ids)
    d = sum(dur[i] for i in ids) / n
    g = sum(ctr[i]["GRBM_GUI_ACTIVE"] for i in ids) / n / 8
    mf = sum(ctr[i]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in ids) / n
    wc = sum(ctr[i]["SQ_WAVE_CYCLES"] for i in ids) / n
    wa = sum(ctr[i]["SQ_WAIT_ANY"] for i in ids) / n
    wi = sum(ctr[i]["SQ_WAIT_INST_ANY"] for i in ids) / n
    ac = sum(ctr[i]["SQ_ACTIVE_INST_ANY"] for i in ids) / n
    print(f"cfg {c:>3}: {d/1000:7.1f} us  cycles {g/1e3:7.1f}k  clock {g/d:5.2f} GHz  MfmaUtil {mf/(g*1024)*100:5.1f}%  wave: wait_any {wa/wc*100:4.1f}% wait_inst {wi/wc*100:4.1f}% active {ac/wc*100:4.1f}%")
PY
