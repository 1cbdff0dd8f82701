"""Per-shape comparison of two `bench.py --dump-ops` files (isolated duration of every launch of one step): which op families got
faster or slower between two trees.  usage: python tools/ops_compare.py OLD.txt NEW.txt [substring of the op name ...]
(OLD may be `REV:path` to read the file from a git revision).  Round 5: this is what showed that a rewrite which made the two
113 MB sums of layer1 10 us faster made the 66 small sums of a step 2 us slower each."""
import collections, re, subprocess, sys


def text(spec):
    if ":" in spec and not spec.startswith("/") and not spec.startswith("."):
        return subprocess.run(["git", "show", spec], capture_output=True, text=True, check=True).stdout
    return open(spec).read()


def load(txt, pat):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for l in txt.splitlines():
        m = re.match(r"\s*([\d.]+) us\s+(\S+)\s+(\S+)\s+(.*)", l)
        if not m or pat not in m.group(2):
            continue
        kv = dict(re.findall(r"(\w+)=(\S+)", m.group(4)))
        key = (m.group(2).replace("stl_", ""), kv.get("H") or kv.get("Hi"), kv.get("W") or kv.get("Wi"), kv.get("C") or kv.get("Ci"), kv.get("Co", "-"),
               kv.get("ks", "-"), kv.get("stride", "-"), kv.get("stuff", "-"), kv.get("nterms") or kv.get("ngrads") or kv.get("shift") or kv.get("n") or "-")
        agg[key][0] += 1
        agg[key][1] += float(m.group(1))
    return agg


old, new = text(sys.argv[1]), text(sys.argv[2])
for pat in (sys.argv[3:] or [""]):
    a, b = load(old, pat), load(new, pat)
    to = tn = 0.0
    for k in sorted(set(a) | set(b), key=lambda k: -(b.get(k, [0, 0])[1])):
        o, n = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
        to, tn = to + o[1], tn + n[1]
        print(f"{' '.join(map(str, k)):56s} n={n[0]:3d}  old {o[1] / max(o[0], 1):7.1f} us  new {n[1] / max(n[0], 1):7.1f} us  sum {n[1] - o[1]:+8.0f} us")
    print(f"[{pat or 'all'}] summed: old {to:.0f} us, new {tn:.0f} us")
