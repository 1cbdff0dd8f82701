"""Parse a rocprofv3 --pmc counter_collection.csv: per kernel name, average counters per dispatch and per wave."""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("conv_core", "conv_ws", "wgrad", "conv1x1")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in agg.items():
    d = max(n[k], 1); w = c.get("SQ_WAVES", 0) / d
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)[:70]
    print(short, f"launches {d} waves/launch {w:.0f}", " ".join(f"{cn}/wave={v / d / max(w, 1):.0f}" for cn, v in c.items() if cn != "SQ_WAVES"))
