"""Which kernels run ALONE (or with one neighbour) in a step?  rocprofv3 kernel-trace CSV (.gz) -> per kernel family,
the milliseconds of the last full step during which exactly 1 / 2 / >= 3 kernels were resident, plus the timeline of the
step cut into its phases (first / last launch of each family).  Speed-ups of kernels that run alone carry over 1:1."""
import csv, gzip, sys, re, collections
f = sys.argv[1]
op = gzip.open if f.endswith('.gz') else open
rows = []
for line in op(f, 'rt'):
    if not line.startswith('"KERNEL_DISPATCH"'):
        continue
    m = re.match(r'"KERNEL_DISPATCH","[^"]*",(\d+),(\d+),(\d+),(\d+),(\d+),"(.*?)",(\d+),(\d{12,}),(\d{12,})(.*)$', line.strip())
    if m:
        rest = [x.strip('"') for x in m.group(10).split(',') if x != ''] + ['0'] * 9
        rows.append({'q': m.group(1), 'n': m.group(6), 's': int(m.group(8)), 'e': int(m.group(9)), 'grid': rest[8], 'wg': rest[5]})
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['n']]
win = rows[adam[-2] + 1:adam[-1] + 1]
t0 = win[0]['s']
def short(n):
    m = re.search(r'conv_core_kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', n)
    if m: return f"conv<k{m.group(2)},WM{m.group(3)}xWN{m.group(4)},MT{m.group(5)},nva{m.group(7)},q{m.group(8)}>"
    for k in ('wgrad64', 'wgrad', 'conv_ws', 'conv1x1', 'conv_r2', 'conv_core', 'fuse_bwd', 'fuse_fwd', 'fuse_flat_big', 'upsample_bwd', 'reduce_slabs', 'weight_prep', 'adam', 'head_bwd', 'head_fwd', 'patch', 'mse', 'bn_running', 'bn_param'):
        if k in n: return k
    return n[:40]
ev = []
for i, r in enumerate(win):
    ev.append((r['s'], 1, i)); ev.append((r['e'], -1, i))
ev.sort()
live = set(); last = t0
alone = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for t, d, i in ev:
    dt = t - last
    if dt > 0 and live:
        c = min(len(live), 3) - 1
        for j in live: alone[short(win[j]['n'])][c] += dt / len(live) if c else dt
    last = t
    if d > 0: live.add(i)
    else: live.discard(i)
print(f"{'family':40s} {'alone ms':>9s} {'with 1':>9s} {'with >=2':>9s}   (shared time split evenly among the co-resident kernels)")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1][0])[:20]:
    print(f"{k:40s} {v[0] / 1e6:9.3f} {v[1] / 1e6:9.3f} {v[2] / 1e6:9.3f}")
print("total alone %.3f ms" % (sum(v[0] for v in alone.values()) / 1e6))
if len(sys.argv) > 2:   # timeline of kernels that ran alone for more than N us
    thr = float(sys.argv[2]) * 1e3
    live = set(); last = t0; acc = collections.Counter()
    for t, d, i in ev:
        if len(live) == 1 and t > last: acc[next(iter(live))] += t - last
        last = t
        if d > 0: live.add(i)
        else: live.discard(i)
    for i, a in sorted(acc.items()):
        if a >= thr:
            r = win[i]
            print(f"  t={(r['s'] - t0) / 1e3:9.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f}  alone {a / 1e3:7.1f}  q{r['q']} grid {r['grid']:>8s} wg {r['wg']:>4s}  {short(r['n'])}")
