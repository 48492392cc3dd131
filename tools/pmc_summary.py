"""Summarise rocprofv3 --pmc csv output (counter_collection.csv): per kernel name, mean of each counter per dispatch.
   python tools/pmc_summary.py <dir> [-k substring]"""
import csv, glob, os, sys, collections, argparse
ap = argparse.ArgumentParser(); ap.add_argument("path"); ap.add_argument("-k", default=""); a = ap.parse_args()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(a.path, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if a.k and a.k not in k: continue
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
