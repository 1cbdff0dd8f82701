"""Offline model of one train step: the planner's op lists (built on the CPU, no kernels run) joined with the measured
one-stream kernel durations (tools/timeline.py window dump of the serial trace), then
  * the dependency-only critical path (infinite queues, zero launch / sync cost),
  * an in-order simulation of the plan's stream assignment (no interference, `sync_us` per cross-stream wait).
usage: python tools/dag_model.py <serial_window.txt> [sync_us]"""
import collections, os, re, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stlpose_amd import capi
from stlpose_amd.arch import ARCHS, registry
from stlpose_amd.engine import Engine, ParamStore

FAM = {"stl_conv_forward": ("conv", "conv_ws", "conv1x1", "conv_core"), "stl_conv_wgrad": ("wgrad", "wgrad64"), "stl_fuse_forward": ("fuse_fwd",),
       "stl_fuse_backward": ("fuse_bwd",), "stl_upsample_backward": ("upsample_bwd",), "stl_patch3x3": ("patch",), "stl_head_forward": ("head_fwd",),
       "stl_head_backward": ("head_bwd",), "stl_reduce_slabs_range": ("reduce_slabs",), "stl_bn_grads_range": ("bn_param",), "stl_conv_wgrad_group": ("wgrad",)}


def load_window(path):
    rows = []
    for line in open(path):
        f = line.split()
        rows.append((float(f[1]), f[-1]))
    return rows


def main():
    win = load_window(sys.argv[1])
    sync = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    reg = registry(ARCHS["w32"])
    eng = Engine(ARCHS["w32"], ParamStore(reg, torch.device("cpu")), 32, 384, 288, capi.BF16, True)
    prog = [k for k in win if not any(s in k[1] for s in ("weight_prep", "FillFunctor", "bn_running", "mse", "sum_partials", "adam", "inc_step"))]
    ops = list(eng.fwd_ops) + list(eng.bwd_ops)
    assert len(prog) == len(ops), (len(prog), len(ops))
    dur = []
    for (d, name), op in zip(prog, ops):
        assert any(name.startswith(f) for f in FAM[op[0]]), (name, op[0])
        dur.append(d)
    other = sum(d for d, n in win) - sum(dur)
    for label, lo, hi in (("forward", 0, len(eng.fwd_ops)), ("backward", len(eng.fwd_ops), len(ops))):
        sub = ops[lo:hi]
        d = dur[lo:hi]
        last, fin_inf = {}, []
        sfin = collections.defaultdict(float)   # per-stream finish time (in-order simulation)
        fin_sim = []
        busy = collections.defaultdict(float)
        chain = 0.0
        for i, (name, desc, strm, reads, writes) in enumerate(sub):
            deps = [last[r] for r in reads if r in last]
            t_inf = max([fin_inf[j] for j in deps], default=0.0) + d[i]
            fin_inf.append(t_inf)
            start = sfin[strm]
            for j in deps:
                if sub[j][2] != strm:
                    start = max(start, fin_sim[j] + sync)
                else:
                    start = max(start, fin_sim[j])
            fin_sim.append(start + d[i])
            sfin[strm] = start + d[i]
            busy[strm] += d[i]
            for w in writes:
                last[w] = i
        off = sum(x for x, o in zip(d, sub) if o[0] in ("stl_conv_wgrad", "stl_conv_wgrad_group", "stl_reduce_slabs_range", "stl_bn_grads_range"))
        print(f"{label}: {len(sub)} ops, serial sum {sum(d) / 1e3:.2f} ms (off-chain {off / 1e3:.2f}), dependency-only critical path {max(fin_inf) / 1e3:.2f} ms, "
              f"in-order {len(sfin)}-stream simulation {max(fin_sim) / 1e3:.2f} ms (sync {sync} us); per-stream busy {[round(busy[s] / 1e3, 2) for s in sorted(busy)]}")
    print(f"outside the programs (weight_prep, fills, loss, optimiser): {other / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
