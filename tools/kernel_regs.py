"""VGPR / SGPR / LDS / spill table of every kernel in a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage x.hip -o x.s 2> x.remarks
       python tools/kernel_regs.py x.remarks [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: (int(re.search(k + r": (\d+)", b).group(1)) if re.search(k + r": (\d+)", b) else -1)
    rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g("VGPR Spill")))
def demangle(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1)); base = n[m.end():m.end() + ln]; rest = n[m.end() + ln:]
    if not rest.startswith("I"):
        return base
    args = []
    for a in re.findall(r"DF16b|Lin\d+E|Li\d+E|Lb[01]E|f", rest[1:rest.index("EEv") + 1] if "EEv" in rest else rest[1:]):
        args.append("bf16" if a == "DF16b" else "f32" if a == "f" else ("-" + a[3:-1]) if a.startswith("Lin") else a[2:-1])
    return base + "<" + ",".join(args) + ">"
names = [demangle(r[0]) for r in rows]
print("vgpr agpr sgpr scratch occ spill  kernel")
for r, d in zip(rows, names):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    if key in d:
        print("%4d %4d %4d %7d %3d %5d  %s" % (r[1], r[2], r[3], r[4], r[5], r[6], d[:170]))
