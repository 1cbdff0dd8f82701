"""Fabric traffic of ONE steady-state train step per kernel family, from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE collected separately, --kernel-trace only).  The step is the window between the
last two `adam_kernel` dispatches of each pass (Dispatch_Id order), so initialisation copies and the isolated
dominant-kernel launches bench.py appends are excluded.  Corrections per MI355X_MICROARCH.md (HBM): counter unit
KB; gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is.

usage: python tools/step_traffic.py <fetch_dir> <write_dir> [out.txt]
"""
import collections, csv, glob, os, re, sys

FAMS = [("wgrad64", "wgrad"), ("wgrad_kernel", "wgrad"), ("conv1x1", "conv1x1"), ("conv_ws", "conv_ws"), ("conv_r2", "conv_r2"), ("conv_core", "conv_core"),
        ("fuse_fwd", "fuse_fwd"), ("fuse_flat_big", "fuse_fwd"), ("fuse_bwd", "fuse_bwd"), ("upsample_bwd", "upsample"), ("reduce_slabs", "reduce_slabs"),
        ("weight_prep", "weight_prep"), ("adam", "adam"), ("head_", "head"), ("patch", "patch"), ("mse", "mse")]


def fam(name):
    for key, f in FAMS:
        if key in name:
            return f
    return "other"


def window(d, counter):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    assert len(adam) >= 2, "need two optimiser launches to window one step"
    win = rows[adam[-2] + 1: adam[-1] + 1]
    agg = collections.defaultdict(float)
    for r in win:
        agg[fam(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
    return agg, len(win)


def main():
    fa, nf = window(sys.argv[1], "FETCH_SIZE")
    wa, nw = window(sys.argv[2], "WRITE_SIZE")
    lines = []
    tot = 0.0
    for k in sorted(set(fa) | set(wa), key=lambda k: -(2 * fa.get(k, 0) + wa.get(k, 0))):
        f, w = 2.0 * fa.get(k, 0.0), wa.get(k, 0.0)
        tot += f + w
        lines.append(f"{k:14s} fetch {f / 1e9:6.2f} GB  write {w / 1e9:6.2f} GB")
    lines.append(f"total {tot / 1e9:.1f} GB/step (2 x FETCH_SIZE + WRITE_SIZE; one steady-state step = {nf} kernels; separate --pmc passes)")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
