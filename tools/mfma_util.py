"""MFMA utilisation per kernel family from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE (own pass, --kernel-trace only).  One steady-state step (between the last two adam launches).
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
counter sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back).  Counter semantics: busy cycles summed over
the SIMDs (16 per v_mfma_f32_16x16x32_bf16).

usage: python tools/mfma_util.py <pmc_dir> [out.json]
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_traffic import fam


def main():
    f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(dict)
    for r in rows:
        per[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"]
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)
    adam = [i for i in ids if "adam_kernel" in per[i]["name"]]
    assert len(adam) >= 2
    win = [i for i in ids if adam[-2] < i <= adam[-1]]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for i in win:
        d = per[i]
        k = fam(d["name"])
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            agg[k][c] += d.get(c, 0.0)
        agg[k]["n"] += 1
    out = {}
    tot_m = tot_c = 0.0
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0) if cyc else 0.0
        tot_m += c["SQ_VALU_MFMA_BUSY_CYCLES"]
        tot_c += cyc
        out[k] = dict(launches=int(c["n"]), kernel_cycles=cyc, mfma_busy_cycles=c["SQ_VALU_MFMA_BUSY_CYCLES"],
                      sq_busy_cycles=c["SQ_BUSY_CYCLES"], mfma_util=round(util, 4))
        print(f"{k:14s} n {int(c['n']):4d}  kernel Mcycles {cyc / 1e6:8.2f}  MFMA util {100 * util:5.1f} %")
    out["_step"] = dict(mfma_busy_cycles=tot_m, sum_kernel_cycles=tot_c, mfma_util_over_kernel_time=round(tot_m / (tot_c * 1024.0), 4),
                        note="profiled passes serialise kernels; utilisation is relative to the summed kernel time")
    print(f"step: MFMA busy / (sum of kernel cycles x 1024 SIMDs) = {100 * tot_m / (tot_c * 1024.0):.1f} %")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
