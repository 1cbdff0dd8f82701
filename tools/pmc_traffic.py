"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

bench.py ends with one isolated launch of every 3x3 stride-1 BN-backward weight-gradient of the
step (time_dominant_kernel), so the LAST `n` wgrad dispatches of each pass are exactly the launches
the roofline entry is about.  Corrections per MI355X_MICROARCH.md (HBM section): counter unit is
KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <n_launches> <out.json>
"""
import csv, glob, json, os, sys


def last_wgrads(d, counter, n):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and ("wgrad_kernel" in r["Kernel_Name"] or "wgrad_ws_kernel" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-n:]
    assert len(rows) == n, (len(rows), n)
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fd, wd, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch_kb = last_wgrads(fd, "FETCH_SIZE", n)
    write_kb = last_wgrads(wd, "WRITE_SIZE", n)
    fetch_b = 2.0 * 1024.0 * sum(fetch_kb) / n   # gfx950 correction: x2
    write_b = 1024.0 * sum(write_kb) / n
    res = dict(kernel="wgrad_kernel / wgrad_ws_kernel <bf16,KS=3,GQ=1> (3x3 stride-1, BN-backward on load)", launches=n,
               fetch_size_raw_kb_per_launch=sum(fetch_kb) / n, write_size_raw_kb_per_launch=sum(write_kb) / n,
               fetch_bytes_per_launch=fetch_b, write_bytes_per_launch=write_b, traffic_bytes_per_launch=fetch_b + write_b,
               fused_bwd=os.environ.get("STLPOSE_FUSED_BWD", "0"), wgrad_group=os.environ.get("STLPOSE_WGRAD_GROUP", "4"),
               note="FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as is; separate --pmc passes")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
