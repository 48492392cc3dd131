#!/bin/bash
# HBM traffic of the GEMM kernel inside the benchmark command, two separate PMC passes (FETCH_SIZE / WRITE_SIZE):
#   tools/traffic.sh <outdir>      -> <outdir>/traffic.json
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$C -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline) > $R/$OUT/$C.log 2>&1
done
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    kt = glob.glob(f"{out}/{C}/*/*kernel_trace.csv"); cc = glob.glob(f"{out}/{C}/*/*counter_collection.csv")
    if not kt or not cc:
        print(C, "missing"); continue
    rows = list(csv.DictReader(open(kt[0])))
    calib_end = max([int(r["End_Timestamp"]) for r in rows if "amax_kernel" in r["Kernel_Name"] or "calib_update" in r["Kernel_Name"]] or [0])
    eul = [int(r["End_Timestamp"]) for r in rows if "euler_kernel" in r["Kernel_Name"] and int(r["Start_Timestamp"]) > calib_end]
    t1 = eul[-1]
    ids = {r["Dispatch_Id"] for r in rows if ("gemm_pp_kernel" in r["Kernel_Name"] or "gemm_w1_kernel" in r["Kernel_Name"]) and calib_end < int(r["Start_Timestamp"]) < t1}
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(cc[0])) if r["Dispatch_Id"] in ids and r["Counter_Name"] == C]
    res[C] = {"launches": len(vals), "mean_raw": sum(vals) / max(1, len(vals))}
    print(C, res[C])
json.dump(res, open(f"{out}/traffic_raw.json", "w"), indent=1)
PY
