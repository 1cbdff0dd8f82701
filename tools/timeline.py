"""Offline analysis of a rocprofv3 kernel-trace CSV (optionally .gz): isolates the last full train
step (between two adam launches), and reports wall time, per-queue busy time, concurrency histogram
and the per-kernel-family sums inside that window."""
import csv, gzip, sys, re, collections
f = sys.argv[1]
op = gzip.open if f.endswith('.gz') else open
rows = []
for line in op(f, 'rt'):
    if not line.startswith('"KERNEL_DISPATCH"'):
        continue
    # kernel names may contain commas (and the capture may have been cut at a fixed field count):
    # locate the name between the 7th field and the two consecutive timestamps
    m = re.match(r'"KERNEL_DISPATCH","[^"]*",(\d+),(\d+),(\d+),(\d+),(\d+),"(.*?)",(\d+),(\d{12,}),(\d{12,})(.*)$', line.strip())
    if not m:
        continue
    rest = [x.strip('"') for x in m.group(10).split(',') if x != '']
    rest += ['0'] * 9
    rows.append({'Queue_Id': m.group(1), 'Kernel_Name': m.group(6), 's': int(m.group(8)), 'e': int(m.group(9)),
                 'LDS_Block_Size': rest[0], 'VGPR_Count': rest[2], 'Accum_VGPR_Count': rest[3], 'Workgroup_Size_X': rest[5], 'Grid_Size_X': rest[8]})
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
assert len(adam) >= 2, len(adam)
i0, i1 = adam[-2] + 1, adam[-1] + 1
win = rows[i0:i1]
t0, t1 = win[0]['s'], max(r['e'] for r in win)
print(f"step window: {len(win)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, sum of kernel time {sum(r['e'] - r['s'] for r in win) / 1e6:.3f} ms")
def short(n):
    m = re.search(r'conv_core_kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', n)
    if m: return f"conv<k{m.group(2)},WM{m.group(3)}xWN{m.group(4)},MT{m.group(5)},NT{m.group(6)},nva{m.group(7)},q{m.group(8)}>"
    for k in ('wgrad64', 'wgrad', 'conv_ws', 'conv1x1', 'conv_r2', 'conv_core', 'fuse_bwd', 'fuse_fwd', 'fuse_flat_big', 'upsample_bwd', 'reduce_slabs', 'weight_prep', 'adam', 'head_bwd', 'head_fwd', 'patch', 'mse', 'copyBuffer', 'FillFunctor', 'bn_running', 'bn_param', 'bwd3x3', 'bn_final'):
        if k in n: return k
    return n[:40]
# events
ev = []
for r in win:
    ev.append((r['s'], 1)); ev.append((r['e'], -1))
ev.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("concurrency histogram (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
q = collections.defaultdict(float)
for r in win: q[r['Queue_Id']] += r['e'] - r['s']
print("per-queue busy ms:", {k: round(v / 1e6, 2) for k, v in sorted(q.items())})
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += (r['e'] - r['s']) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:44s} n {c:4d}  avg {t / c:8.1f} us  {t / 1e3:7.2f} ms")
# gaps between consecutive kernels of one hardware queue: a host that cannot keep up with the eager replay would
# show as a floor of >= 3-4 us under EVERY launch; dependency waits (branch imbalance, exchange joins) show as a
# few long gaps instead
buckets = [(0, 0.5), (0.5, 2), (2, 5), (5, 10), (10, 50), (50, 1e9)]
gh = collections.Counter(); gt = collections.Counter()
for qid in sorted(q):
    ks = sorted((r for r in win if r['Queue_Id'] == qid), key=lambda r: r['s'])
    last_e = None
    for r in ks:
        if last_e is not None:
            g = max(0.0, (r['s'] - last_e) / 1e3)
            for lo, hi in buckets:
                if lo <= g < hi:
                    gh[(lo, hi)] += 1; gt[(lo, hi)] += g
        last_e = r['e'] if last_e is None else max(last_e, r['e'])
print("same-queue gaps (us): " + ", ".join(f"[{lo},{hi if hi < 1e8 else 'inf'}) n={gh[(lo, hi)]} sum={gt[(lo, hi)] / 1e3:.2f}ms" for lo, hi in buckets))
# phases: forward ends at mse kernel
mse = [r for r in win if 'mse_kernel' in r['Kernel_Name']]
if mse:
    print(f"forward part: {(mse[0]['s'] - t0) / 1e6:.3f} ms, backward+opt part: {(t1 - mse[0]['s']) / 1e6:.3f} ms")
if len(sys.argv) > 2:   # dump the window
    with open(sys.argv[2], 'w') as o:
        for r in win:
            o.write(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['Queue_Id']} g{r['Grid_Size_X']:>8} wg{r['Workgroup_Size_X']:>4} v{r['VGPR_Count']:>3} a{r['Accum_VGPR_Count']:>3} lds{r['LDS_Block_Size']:>6} {short(r['Kernel_Name'])}\n")
