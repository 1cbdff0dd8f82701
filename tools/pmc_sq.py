"""Per kernel family of one steady-state step: SQ wave-state counters (rocprofv3 --pmc, serialised dispatches).
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_traffic import fam


def load(d):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        return {}
    rows = list(csv.DictReader(open(max(fs, key=os.path.getmtime))))
    per = collections.defaultdict(dict)
    for r in rows:
        per[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    adam = [i for i in ids if "adam_kernel" in per[i]["name"]]
    win = [i for i in ids if adam[-2] < i <= adam[-1]]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for i in win:
        n = per[i]["name"]
        k = fam(n)
        if k == "conv_core":
            m = re.search(r"conv_core_kernelI\w+?Li\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELb([01])E", n)   # Q = 8th template argument
            k = "conv_q1" if (m and m.group(1) == "1") else "conv_q0"
        if k == "conv_r2":   # conv_r2_kernel<T, TY, Q, ...>: Q = 3rd template argument (data gradient)
            m = re.search(r"conv_r2_kernelI(?:DF16b|DF16_)(?:DF16b|DF16_|S\d*_)Lb([01])E", n)
            k = "r2_q1" if (m and m.group(1) == "1") else "r2_q0"
        for c, v in per[i].items():
            if c != "name":
                agg[k][c] += v
        agg[k]["n"] += 1
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2]) if len(sys.argv) > 2 else {}
for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"]):
    c = a[k]
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    line = (f"{k:12s} n {int(c['n']):4d} waves/launch {c['SQ_WAVES'] / c['n']:7.0f}  wave-cycles: parked {100 * c['SQ_WAIT_ANY'] / wc:5.1f} %  issue-stall {100 * c['SQ_WAIT_INST_ANY'] / wc:5.1f} %"
            f"  issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:5.1f} % (VALU {100 * c['SQ_ACTIVE_INST_VALU'] / wc:5.1f} %)  VALU insts/wave {c['SQ_INSTS_VALU'] / max(c['SQ_WAVES'], 1):7.0f}  SALU/wave {c['SQ_INSTS_SALU'] / max(c['SQ_WAVES'], 1):6.0f}")
    if k in b:
        d = b[k]
        line += (f"  | LDS {100 * d['SQ_ACTIVE_INST_LDS'] / wc:4.1f} % VMEM {100 * d['SQ_ACTIVE_INST_VMEM'] / wc:4.1f} % SCA {100 * d['SQ_ACTIVE_INST_SCA'] / wc:4.1f} % MISC {100 * d['SQ_ACTIVE_INST_MISC'] / wc:4.1f} %"
                 f"  LDS insts/wave {d['SQ_INSTS_LDS'] / max(c['SQ_WAVES'], 1):5.0f} VMEM rd/wr {d['SQ_INSTS_VMEM_RD'] / max(c['SQ_WAVES'], 1):4.0f}/{d['SQ_INSTS_VMEM_WR'] / max(c['SQ_WAVES'], 1):4.0f}")
    print(line)
