"""Replay a WINDOW of the planned backward (or forward) program on its streams and on one stream: what does the overlap of
the four branch chains buy in that window?  usage: python tools/subplan_probe.py [fwd|bwd] [first_op] [n_ops] [nowg]
(GPU box; W32 384x288 bs 32 bf16).  Prints the window's op mix, its time on the plan's streams and on one stream."""
import ctypes as C, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import PoseHighResolutionNet, capi
from stlpose_amd.train_step import TrainStep

which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
nowg = len(sys.argv) > 4 and sys.argv[4] == "nowg"
torch.manual_seed(0)
m = PoseHighResolutionNet("w32", os.environ.get("STLPOSE_DTYPE", "mixed")).cuda()
ts = TrainStep(m, 32, 384, 288)
g = torch.Generator().manual_seed(1)
ts.load_batch(torch.randn(32, 3, 384, 288, generator=g).cuda(), torch.rand(32, 17, 96, 72, generator=g).cuda(), torch.ones(32, 17, 1).cuda())
for _ in range(3):
    ts.step()          # real activations / statistics in every buffer
torch.cuda.synchronize()
eng = ts.eng
ops = list(eng.fwd_ops if which == "fwd" else eng.bwd_ops)[first:first + count]
if nowg:
    ops = [o for o in ops if not o[0].startswith("stl_conv_wgrad") and o[0] not in ("stl_reduce_slabs_range", "stl_bn_grads_range")]
print(f"{which} ops [{first}, {first + count}): {len(ops)} launches", dict(collections.Counter((o[0].replace('stl_', ''), o[2]) for o in ops)))


def run(oplist, reps=20):
    eng._progs.pop(id(oplist), None)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        eng._run(oplist, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng._run(oplist, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


saved = getattr(eng, "buckets", [])
eng.buckets = []          # bucket events refer to the full program
multi = run(ops)
serial_ops = [(n, d, 0, r, w) for n, d, _s, r, w in ops]
one = run(serial_ops)
eng.buckets = saved
print(f"plan streams: {multi:8.1f} us   one stream: {one:8.1f} us   ratio {one / multi:.2f}")
