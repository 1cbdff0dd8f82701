"""MFMA utilisation per kernel family from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE (own pass, --kernel-trace only).  One steady-state step (between the last two adam launches).
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
counter sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back).  Counter semantics: busy cycles summed over
the SIMDs (16 per v_mfma_f32_16x16x32_bf16).

Reconciliation with the flop-based figure: the two differ in BOTH terms of the fraction --
  * numerator: the counter sees every MFMA issued, also those on zero-padded tile rows / virtual rows / ragged channel
    chunks; executed FLOPs = busy cycles / 16 x 16384 (one 16x16x32 bf16 MFMA), printed next to the algorithmic
    FLOPs of the step (103.11 GFLOP x 32 images);
  * denominator: kernel cycles of THIS pass (GRBM_GUI_ACTIVE / 8), which counter collection stretches relative to
    the plain kernel trace; with `serial_ms` (summed kernel time of the un-instrumented one-stream trace,
    profiles/r*_trace_serial_summary.txt) the same busy cycles are also put on that time base at the 2.4 GHz the
    2.5 PFLOP/s peak is quoted at.

usage: python tools/mfma_util.py <pmc_dir> [out.json] [serial_ms]
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
    alg = 103.11e9 * 32
    executed = tot_m / 16.0 * 16384.0
    out["_step"].update(executed_mfma_flops=executed, algorithmic_flops=alg, executed_over_algorithmic=round(executed / alg, 4))
    print(f"executed MFMA FLOPs {executed / 1e12:.3f} T vs algorithmic {alg / 1e12:.3f} T  (x{executed / alg:.3f}: padded tiles / virtual rows)")
    if len(sys.argv) > 3:
        ms = float(sys.argv[3])
        u_cnt = tot_m / (ms * 1e-3 * 2.4e9 * 1024.0)
        u_flop = alg / (ms * 1e-3) / 2.5e15
        out["_step"].update(serial_ms=ms, mfma_util_on_serial_trace=round(u_cnt, 4), flop_based_on_serial_trace=round(u_flop, 4))
        print(f"on the un-instrumented serial trace ({ms:.2f} ms of kernel time, 2.4 GHz): counter-based {100 * u_cnt:.1f} %, flop-based {100 * u_flop:.1f} %;"
              f" ratio {u_cnt / u_flop:.3f} = executed / algorithmic FLOPs x (peak FLOP per SIMD-cycle 2.5e15 / (1024 x 2.4e9) = 1017) / (1024 per MFMA cycle) = {executed / alg * 2.5e15 / (1024 * 2.4e9) / 1024.0:.3f}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
