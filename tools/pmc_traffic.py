"""HBM traffic of the DOMINANT kernel (whichever bench.py measured it to be) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

bench.py ends with one isolated launch of every launch of its dominant kernel family (roofline.kernel, launches_per_step), so the
LAST n dispatches of that instantiation in each pass are exactly the launches the roofline entry is about.  Corrections per
MI355X_MICROARCH.md (HBM section): counter unit is KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads ->
doubled; WRITE_SIZE exact.  The record carries the library's build id: bench.py uses it only for the same kernel sources.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <bench.json> <out.json>
"""
import csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_names import canon  # noqa: E402


def last_launches(d, counter, kernel, n):
    f = max(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and canon(r["Kernel_Name"]) == kernel]
    if not rows:   # names rocprofv3 demangled itself (garbled for bf16 templates): fall back to the base name
        base = kernel.split("<")[0]
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and base in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-n:]
    assert len(rows) == n, (len(rows), n, kernel)
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fd, wd, bench, out = sys.argv[1:5]
    b = json.load(open(bench))
    kernel, n = b["roofline"]["kernel"], int(b["roofline"]["launches_per_step"])
    fetch_kb = last_launches(fd, "FETCH_SIZE", kernel, n)
    write_kb = last_launches(wd, "WRITE_SIZE", kernel, n)
    fetch_b = 2.0 * 1024.0 * sum(fetch_kb) / n   # gfx950 correction: x2
    write_b = 1024.0 * sum(write_kb) / n
    from stlpose_amd import build
    res = dict(kernel=kernel, launches=n, dtype={"f16/bf16": "mixed", "bf16": "bf16", "f32": "fp32"}.get(b["dtype"], b["dtype"]), build_id=build.source_id(),
               fetch_size_raw_kb_per_launch=sum(fetch_kb) / n, write_size_raw_kb_per_launch=sum(write_kb) / n,
               fetch_bytes_per_launch=fetch_b, write_bytes_per_launch=write_b, traffic_bytes_per_launch=fetch_b + write_b,
               algorithmic_bytes_per_launch=b["roofline"]["algorithmic_bytes_per_launch"],
               note="FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as is; separate --pmc passes")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
