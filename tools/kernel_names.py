"""Canonical kernel names: maps the (partly mangled) names of a rocprofv3 kernel_stats / kernel_trace CSV onto the form the library
reports through stl_last_kernel() and bench.py prints -- "conv_core_kernel<bf16,3,4,2,4,2,3,1,0,1,-1,0,2>": element type, then the
template arguments in declaration order (a trailing element-type argument as its STL_* code: 0 f32, 1 bf16, 2 f16).
usage: python tools/kernel_names.py kernel_stats.csv  -> the CSV with a canonical first column, sorted by total time"""
import csv, re, sys


def canon(n: str) -> str:
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        ln = int(m.group(1))
        base, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
        if not rest.startswith("I"):
            return base
        toks = re.findall(r"DF16b|DF16_|S[0-9A-Z]*_|Lin\d+E|Li\d+E|Lb[01]E|f", rest[1:rest.index("EEv") + 1] if "EEv" in rest else rest[1:])
        ty = {"DF16b": ("bf16", "1"), "DF16_": ("f16", "2"), "f": ("f32", "0")}
        args = []
        for i, a in enumerate(toks):
            if a.startswith("S") and a.endswith("_"):   # substitution: the same type as the first template argument
                a = toks[0]
            if a in ty:
                args.append(ty[a][0] if i == 0 else ty[a][1])
            else:
                args.append(("-" + a[3:-1]) if a.startswith("Lin") else a[2:-1])
        return base + "<" + ",".join(args) + ">"
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


if __name__ == "__main__":
    rows = list(csv.reader(open(sys.argv[1])))
    hdr, body = rows[0], rows[1:]
    ti = hdr.index("TotalDurationNs") if "TotalDurationNs" in hdr else 2
    body.sort(key=lambda r: -float(r[ti]))
    w = csv.writer(sys.stdout)
    w.writerow(["Canonical"] + hdr)
    for r in body:
        w.writerow([canon(r[0])] + r)
