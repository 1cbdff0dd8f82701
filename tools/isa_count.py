"""Count instructions per basic block of one kernel in a hipcc -S listing, weighted by a crude
loop-depth guess (blocks that are the target of a backward branch).  usage: isa_count.py file.s <mangled-substring>"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
blocks = []; cur = ["entry", collections.Counter(), []]
for l in body[1:]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), collections.Counter(), []]; continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    cls = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
           "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cur[1][cls] += 1
    if op.startswith("s_cbranch") or op == "s_branch": cur[2].append(t.split()[-1])
blocks.append(cur)
idx = {b[0]: i for i, b in enumerate(blocks)}
tot = collections.Counter()
for i, (name, c, br) in enumerate(blocks):
    back = [t for t in br if t in idx and idx[t] <= i]
    n = sum(c.values())
    if n >= 25 or back:
        print(f"{name:14s} n={n:5d} valu={c['valu']:5d} salu={c['salu']:4d} lds={c['lds']:4d} vmem={c['vmem']:4d} mfma={c['mfma']:4d} {'<-loop to ' + ','.join(back) if back else ''}")
    tot.update(c)
print("total", dict(tot))
