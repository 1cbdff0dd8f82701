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
bw = [r for r in win[hb:] if 'adam' not in r['n'] and 'inc_step' not in r['n'] and 'sum_partials' not in r['n']]   # (the loss scalar's sum is enqueued behind backward since round 5)
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

# Hypothesis check: does a cross-stream wait release when the producer OP ends, or when everything the producer STREAM had been
# given before the consumer was issued has ended?  alt_ready = max(end of the stream predecessor, for every waited stream t the
# end of the LAST op of t with a smaller index than the consumer).
if len(sys.argv) > 3:
    last_before = {}
    n_hit = n_tot = 0; resid_a = resid_b = 0.0
    for i, o in enumerate(ops):
        s = o[2]
        if waits[i]:
            deps_true = [end[w] for w in waits[i]] + ([end[pp]] if (pp := max((j for j in pers[s] if j < i), default=None)) is not None else [])
            alt = list(deps_true)
            for w in waits[i]:
                t = ops[w][2]
                j = max(j for j in pers[t] if j < i)
                alt.append(end[j])
            d_true, d_alt = start[i] - max(deps_true), start[i] - max(alt)
            n_tot += 1
            if d_true > 25:
                n_hit += 1; resid_a += d_true; resid_b += max(d_alt, 0.0)
                print(f"#{i} s{s} start {start[i]:8.1f}  delay vs waited ops {d_true:6.1f}  vs everything queued on their streams before it {d_alt:6.1f}")
    print(f"{n_hit} delayed ops with waits: summed delay {resid_a / 1e3:.2f} ms against the waited ops, {resid_b / 1e3:.2f} ms against the alternative")
