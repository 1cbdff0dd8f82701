import csv, sys, glob, collections, re
d = sys.argv[1]; nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import os; f = max(glob.glob(d + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
def short(n):
    m = re.search(r'conv_core_kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)', n)
    if m: return f"conv<{'bf16' if 'DF16b' in m.group(1) else 'f32'},k{m.group(2)},WM{m.group(3)}xWN{m.group(4)},MT{m.group(5)},NT{m.group(6)},nva{m.group(7)},q{m.group(8)}>"
    m = re.search(r'wgrad_kernelI(\w+?)Li(\d)', n)
    if m: return f"wgrad<{'bf16' if 'DF16b' in m.group(1) else 'f32'},k{m.group(2)}>"
    for k in ('fuse_bwd','fuse_fwd','upsample_bwd','reduce_slabs','weight_prep','adam','head_bwd','head_fwd','patch','mse','copyBuffer','FillFunctor','bn_running','bn_param'):
        if k in n: return k
    return n[:40]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name'])
    agg[k][0] += 1; agg[k][1] += dt; tot += dt
print(f"total kernel time {tot/1e3/nsteps:.2f} ms/step over {nsteps} steps")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:50s} calls/step {c/nsteps:7.1f}  avg {t/c:8.1f} us  {t/1e3/nsteps:7.2f} ms/step  {100*t/tot:5.1f}%")
