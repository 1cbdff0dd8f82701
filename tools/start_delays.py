"""Does a launch start when its dependencies allow?  Joins a rocprofv3 kernel trace of the default step (W32 384x288 bs 32 bf16)
with the planner's backward program (rebuilt on the CPU: same op list, same streams, same cross-stream waits): the kernels of
hardware queue i, in order, are the ops of stream i, in order.  For every op: ready = max(end of its stream predecessor, ends of the
ops it waits for); delay = start - ready.  A delay is time the launch spent queued although nothing it depends on was running.

usage: python tools/start_delays.py <trace.csv[.gz]> [min_delay_us]"""
import gzip, re, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import PoseHighResolutionNet, capi
from stlpose_amd.engine import Engine
f = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
m = PoseHighResolutionNet("w32", "bf16"); m._pack(torch.device("cpu"))
e = Engine(m.arch, m._store, 32, 384, 288, capi.BF16, True)
ops = e.bwd_ops; waits, _ = e._schedule(ops)
rows = []
for line in (gzip.open(f, 'rt') if f.endswith('.gz') else open(f)):
    if not line.startswith('"KERNEL_DISPATCH"'): continue
    mm = re.match(r'"KERNEL_DISPATCH","[^"]*",(\d+),(\d+),(\d+),(\d+),(\d+),"(.*?)",(\d+),(\d{12,}),(\d{12,})(.*)$', line.strip())
    if mm: rows.append({'q': mm.group(1), 'n': mm.group(6), 's': int(mm.group(8)), 'e': int(mm.group(9))})
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['n']]
win = rows[adam[-2] + 1:adam[-1] + 1]; t0 = win[0]['s']
hb = [i for i, r in enumerate(win) if 'head_bwd' in r['n']][0]
bw = [r for r in win[hb:] if 'adam' not in r['n'] and 'inc_step' not in r['n']]
perq = collections.defaultdict(list)
for r in bw: perq[r['q']].append(r)
pers = collections.defaultdict(list)
for i, o in enumerate(ops): pers[o[2]].append(i)
# queue <-> stream by launch count
qmap = {}
for q, lst in perq.items():
    cand = [s for s, v in pers.items() if len(v) == len(lst) and s not in qmap.values()]
    assert cand, f"queue {q}: {len(lst)} kernels match no stream ({ {s: len(v) for s, v in pers.items()} })"
    qmap[q] = cand[0]
start, end = {}, {}
for q, lst in perq.items():
    for r, i in zip(lst, pers[qmap[q]]): start[i] = (r['s'] - t0) / 1e3; end[i] = (r['e'] - t0) / 1e3
def nm(i):
    n, d = ops[i][0], ops[i][1]; x = ''
    if n == 'stl_conv_forward': x = f"Ci{d.Ci} Co{d.Co} k{d.ks} {d.Hi}x{d.Wi}"
    elif n == 'stl_fuse_backward': x = f"C{d.C} {d.H}x{d.W}"
    elif n == 'stl_conv_wgrad': x = f"Ci{d.Ci} Co{d.Co} k{d.ks} s{d.stride} {d.Hi}x{d.Wi}"
    elif n == 'stl_conv_wgrad_group': x = f"n{d.n} Ci{d.members[0].Ci}"
    return f"#{i} s{ops[i][2]} {n[4:]} {x}"
prev = {}; tot = 0.0; big = []; byk = collections.Counter()
for i, o in enumerate(ops):
    s = o[2]
    deps = [end[w] for w in waits[i]] + ([end[prev[s]]] if s in prev else [])
    d = start[i] - (max(deps) if deps else start[i])
    tot += max(d, 0.0); byk[o[0][4:]] += max(d, 0.0)
    if d > thr: big.append((d, i))
    prev[s] = i
print(f"backward program: {len(ops)} ops on {len(pers)} streams; sum of start delays beyond known dependencies {tot / 1e3:.2f} ms "
      f"(backward wall {(max(end.values()) - min(start.values())) / 1e3:.2f} ms)")
print("by op kind (ms):", {k: round(v / 1e3, 2) for k, v in byk.most_common()})
for d, i in big:
    print(f"delay {d:6.1f} us  start {start[i]:8.1f}  {nm(i)}  waits {[(w, 's%d' % ops[w][2], round(end[w], 1)) for w in waits[i]]}")
