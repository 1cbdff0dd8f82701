"""Instruction-cache counters per kernel family from a rocprofv3 --pmc pass (SQC_ICACHE_REQ SQC_ICACHE_HITS
SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE), one steady-state step.  usage: python tools/icache_stats.py <pmc_dir>"""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_traffic import fam
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
adam = [i for i in ids if "adam_kernel" in per[i]["name"]]
win = [i for i in ids if adam[-2] < i <= adam[-1]]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = sorted({c for i in win for c in per[i] if c != "name"})
for i in win:
    k = fam(per[i]["name"])
    for c in names: agg[k][c] += per[i].get(c, 0.0)
    agg[k]["n"] += 1
print("family         n   " + "  ".join(f"{c[-18:]:>18s}" for c in names) + "   (per launch)")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0)):
    print(f"{k:14s} {int(c['n']):4d} " + "  ".join(f"{c[x] / c['n']:18.0f}" for x in names))
