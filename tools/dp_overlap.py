"""Per gradient bucket: how long after the bucket is final does its all-reduce kernel start?  

Input: a rocprofv3 --kernel-trace CSV (optionally .gz) of `STLPOSE_DP_FORCE=1 python bench.py ...` (one-rank RCCL rehearsal: the
bucketed all-reduce is issued for real on a one-rank communicator).  A bucket is final when its `bn_param_grads_kernel` launch
ends (the op whose event the communication stream waits for, train_step._allreduce); the collective is the next RCCL kernel
(ncclDevKernel* / *AllReduce*) that starts after it.  With GPU_MAX_HW_QUEUES=4 RCCL's stream shares a hardware queue with one
of the four compute streams, so the gap shows whether the collective sits in-order behind compute kernels of that queue.
usage: python tools/dp_overlap.py trace.csv[.gz] [out.txt]
"""
import gzip, re, sys

f = sys.argv[1]
op = gzip.open if f.endswith(".gz") else open
rows = []
for line in op(f, "rt"):
    if not line.startswith('"KERNEL_DISPATCH"'):
        continue
    m = re.match(r'"KERNEL_DISPATCH","[^"]*",(\d+),(\d+),(\d+),(\d+),(\d+),"(.*?)",(\d+),(\d{12,}),(\d{12,})', line.strip())
    if m:
        rows.append(dict(q=m.group(1), name=m.group(6), s=int(m.group(8)), e=int(m.group(9))))
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["name"]]
assert len(adam) >= 2, "need two optimiser launches to window one step"
win = rows[adam[-2] + 1: adam[-1] + 1]
t0 = win[0]["s"]
is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower() or "allreduce" in n.lower()   # noqa: E731
colls = [r for r in win if is_coll(r["name"])]
buckets = [r for r in win if "bn_param_grads" in r["name"]]
lines = [f"step window: {len(win)} kernels, wall {(max(r['e'] for r in win) - t0) / 1e6:.3f} ms; {len(buckets)} gradient buckets, {len(colls)} collective kernels",
         "bucket  final at (ms)  collective start (ms)  gap (us)  collective (us)  queue(bucket/coll)  kernels on the collective's queue in the gap"]
used = set()
for i, b in enumerate(buckets):
    c = next((r for r in colls if r["s"] >= b["e"] and id(r) not in used), None)
    if c is None:
        lines.append(f"{i:6d}  {(b['e'] - t0) / 1e6:13.3f}  (no collective kernel after it)")
        continue
    used.add(id(c))
    between = [r for r in win if r["q"] == c["q"] and r is not c and r["s"] < c["s"] and r["e"] > b["e"]]
    lines.append(f"{i:6d}  {(b['e'] - t0) / 1e6:13.3f}  {(c['s'] - t0) / 1e6:21.3f}  {(c['s'] - b['e']) / 1e3:8.1f}  {(c['e'] - c['s']) / 1e3:15.1f}  {b['q']:>6s}/{c['q']:<6s}  "
                 f"{len(between)} ({sum(min(r['e'], c['s']) - max(r['s'], b['e']) for r in between) / 1e3:.1f} us busy)")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
