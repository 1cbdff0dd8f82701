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
is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower() or "allreduce" in n.lower()   # noqa: E731
# A ONE-rank communicator launches no kernel for an all-reduce.  With STLPOSE_BF16_BUCKETS=1 the communication stream still
# carries, per bucket and behind the same event, the fp32 -> bf16 staging copy of the bucket (dp.FlatAllReduce._send): that
# kernel stands in for the collective's start.
is_stand_in = lambda n: "bfloat16_copy_kernel" in n   # noqa: E731
win, colls, stand_in = None, [], False
for k in range(len(adam) - 1, 0, -1):   # the LAST step that carries collectives (bench.py ends with steps that have them switched off)
    w = rows[adam[k - 1] + 1: adam[k] + 1]
    c = [r for r in w if is_coll(r["name"])]
    if not c:
        c = [r for r in w if is_stand_in(r["name"])]
        if c:
            stand_in = True
    nb = sum(1 for r in w if "bn_param_grads" in r["name"])
    if c and len(c) == nb:   # exactly one per bucket: a real step (bench.py's per-bucket timing loop issues three per bucket)
        win, colls = w, c
        break
assert win is not None, "no step with a collective (or its stand-in) in the trace: run with STLPOSE_DP_FORCE=1 [STLPOSE_BF16_BUCKETS=1]"
t0 = win[0]["s"]
buckets = [r for r in win if "bn_param_grads" in r["name"]]
lines = ["(no RCCL kernel in the trace: one-rank communicator; the bucket's bf16 staging copy on the communication stream stands in for the collective)"] if stand_in else []
lines += [f"step window: {len(win)} kernels, wall {(max(r['e'] for r in win) - t0) / 1e6:.3f} ms; {len(buckets)} gradient buckets, {len(colls)} collective kernels",
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
