"""Who waits for whom?  rocprofv3 kernel-trace CSV (.gz) -> for every idle gap of a hardware queue longer than N us inside the
last full step: the kernel that started after the gap and the kernel on ANOTHER queue that ended last before that start
(the launch the gap was most likely waiting for).  Summed per releasing kernel family: where the critical chains are."""
import gzip, re, sys, collections
f = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = []
for line in (gzip.open(f, 'rt') if f.endswith('.gz') else open(f)):
    if not line.startswith('"KERNEL_DISPATCH"'): continue
    m = re.match(r'"KERNEL_DISPATCH","[^"]*",(\d+),(\d+),(\d+),(\d+),(\d+),"(.*?)",(\d+),(\d{12,}),(\d{12,})(.*)$', line.strip())
    if m:
        rest = [x.strip('"') for x in m.group(10).split(',') if x != ''] + ['0'] * 9
        rows.append({'q': m.group(1), 'n': m.group(6), 's': int(m.group(8)), 'e': int(m.group(9)), 'blocks': int(rest[8]) // max(1, int(rest[5]))})
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['n']]
win = rows[adam[-2] + 1:adam[-1] + 1]; t0 = win[0]['s']
def short(n):
    m = re.search(r'conv_core_kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', n)
    if m: return f"conv<WM{m.group(3)}xWN{m.group(4)},MT{m.group(5)},q{m.group(8)}>"
    for k in ('wgrad64', 'wgrad_ws', 'wgrad', 'conv_ws', 'conv1x1', 'conv_r2', 'conv_core', 'fuse_bwd', 'fuse_fwd', 'upsample_bwd', 'reduce_slabs', 'adam', 'head_bwd', 'head_fwd', 'mse'):
        if k in n: return k
    return n[:30]
last = {}; tot = collections.Counter(); cnt = collections.Counter()
out = []
for r in win:
    q = r['q']
    if q in last:
        gap = (r['s'] - last[q]) / 1e3
        if gap >= thr:
            cands = [o for o in win if o['q'] != q and o['e'] <= r['s'] + 2000 and o['e'] > last[q]]
            rel = max(cands, key=lambda o: o['e']) if cands else None
            key = (short(rel['n']) + f" [{rel['blocks']} blk]") if rel else "-"
            tot[key] += gap; cnt[key] += 1
            out.append(f"t={(r['s'] - t0) / 1e3:8.1f} q{q} idle {gap:6.1f} us -> {short(r['n'])}  released by q{rel['q'] if rel else '-'} {key} dur {((rel['e'] - rel['s']) / 1e3) if rel else 0:.1f}")
    last[q] = r['e']
print("idle queue time by releasing kernel (gaps >= %.0f us):" % thr)
for k, v in tot.most_common(25): print(f"  {v / 1e3:7.3f} ms  n {cnt[k]:3d}  {k}")
if len(sys.argv) > 3: print("\n".join(out))
