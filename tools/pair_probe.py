"""Judge a conv kernel IN COMPANY, not alone: launches of the benchmarked plan (W32, 384x288, batch 32, mixed
mode) are replayed alone on one stream and in pairs on two streams -- a C >= 64 data gradient / forward conv beside the grouped
3x3 weight gradient, beside the C = 32 conv, beside a copy of itself -- with HIP events around N back-to-back launches per stream.
Run once per kernel generation: `STL_CONV_R2=0 python tools/pair_probe.py` (one-per-CU block) and `python tools/pair_probe.py`."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stlpose_amd import PoseHighResolutionNet, capi  # noqa: E402
from stlpose_amd.train_step import TrainStep  # noqa: E402

REPS = int(os.environ.get("PAIR_REPS", "40"))


def pick(ops, pred):
    return next(o for o in ops if pred(o))


def run(op, stream, n):
    name, desc = op[0], op[1]
    fn = getattr(capi.lib(), name)
    for _ in range(n):
        rc = fn(C.byref(desc), stream.cuda_stream)
        assert rc == 0, capi.lib().stl_last_error().decode()


def time_alone(op, n=REPS):
    s = torch.cuda.current_stream()
    run(op, s, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    run(op, s, n)
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def time_pair(a, b, sa, sb, n=REPS):
    """a on stream sa, b on stream sb, n launches each, started together; per-launch time of each stream's run"""
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    go = torch.cuda.Event()
    go.record(torch.cuda.current_stream())
    sa.wait_event(go), sb.wait_event(go)
    ev[0].record(sa), ev[2].record(sb)
    # interleave the host's issue so that neither queue runs dry
    fa, fb = getattr(capi.lib(), a[0]), getattr(capi.lib(), b[0])
    for _ in range(n):
        assert fa(C.byref(a[1]), sa.cuda_stream) == 0 and fb(C.byref(b[1]), sb.cuda_stream) == 0
    ev[1].record(sa), ev[3].record(sb)
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n, ev[2].elapsed_time(ev[3]) * 1e3 / n


def main():
    torch.manual_seed(0)
    m = PoseHighResolutionNet("w32", "mixed").cuda()
    ts = TrainStep(m, 32, 384, 288, optimizer="adam", lr=1e-3)
    g = torch.Generator().manual_seed(1)
    ts.load_batch(torch.randn(32, 3, 384, 288, generator=g).cuda(), torch.rand(32, 17, 96, 72, generator=g).cuda(), torch.ones(32, 17, 1).cuda())
    for _ in range(2):
        ts.step()
    torch.cuda.synchronize()
    e = ts.eng
    conv = lambda o: o[0] == "stl_conv_forward"   # noqa: E731
    ops = {
        "dgrad C=64 48x36 (mask_y)": pick(e.bwd_ops, lambda o: conv(o) and o[1].Ci == 64 and o[1].Co == 64 and o[1].Hi == 48 and o[1].ks == 3 and o[1].mask_y and not o[1].mask_z and not o[1].stuff),
        "dgrad C=64 48x36 (mask_y + addend + mask_z)": pick(e.bwd_ops, lambda o: conv(o) and o[1].Ci == 64 and o[1].Co == 64 and o[1].Hi == 48 and o[1].ks == 3 and o[1].mask_z),
        "dgrad C=128 24x18 (mask_y)": pick(e.bwd_ops, lambda o: conv(o) and o[1].Ci == 128 and o[1].Co == 128 and o[1].ks == 3 and o[1].mask_y and not o[1].mask_z and not o[1].stuff),
        "forward C=64 48x36": pick(e.fwd_ops, lambda o: conv(o) and o[1].Ci == 64 and o[1].Co == 64 and o[1].Hi == 48 and o[1].ks == 3 and o[1].stride == 1 and o[1].src.mode == capi.SRC_BN),
        "forward C=128 24x18": pick(e.fwd_ops, lambda o: conv(o) and o[1].Ci == 128 and o[1].Co == 128 and o[1].ks == 3 and o[1].stride == 1 and o[1].src.mode == capi.SRC_BN),
        "forward C=256 12x9": pick(e.fwd_ops, lambda o: conv(o) and o[1].Ci == 256 and o[1].Co == 256 and o[1].ks == 3 and o[1].stride == 1 and o[1].src.mode == capi.SRC_BN),
        "dgrad C=256 12x9 (mask_y)": pick(e.bwd_ops, lambda o: conv(o) and o[1].Ci == 256 and o[1].Co == 256 and o[1].ks == 3 and o[1].mask_y and not o[1].mask_z and not o[1].stuff),
        "dgrad C=32 96x72 (mask_y)": pick(e.bwd_ops, lambda o: conv(o) and o[1].Ci == 32 and o[1].Co == 32 and o[1].Hi == 96 and o[1].ks == 3 and o[1].mask_y and not o[1].mask_z and not o[1].stuff),
        "forward C=32 96x72": pick(e.fwd_ops, lambda o: conv(o) and o[1].Ci == 32 and o[1].Co == 32 and o[1].Hi == 96 and o[1].ks == 3 and o[1].stride == 1 and o[1].src.mode == capi.SRC_BN),
        "wgrad group C=32 96x72": pick(e.bwd_ops, lambda o: o[0] == "stl_conv_wgrad_group" and o[1].members[0].Ci == 32 and o[1].members[0].Hi == 96),
        "wgrad group C=64 48x36": pick(e.bwd_ops, lambda o: o[0] == "stl_conv_wgrad_group" and o[1].members[0].Ci == 64 and o[1].members[0].Hi == 48),
    }
    print(f"STL_CONV_R2={os.environ.get('STL_CONV_R2', '(default 3)')}  build {capi.lib().stl_build_id().decode()}  {REPS} launches per measurement")
    alone = {}
    for k, o in ops.items():
        alone[k] = time_alone(o)
        fn = getattr(capi.lib(), o[0])
        fn(C.byref(o[1]), torch.cuda.current_stream().cuda_stream)
        print(f"alone  {k:46s} {alone[k]:7.2f} us   [{capi.lib().stl_last_kernel().decode()}]")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    pairs = [("dgrad C=64 48x36 (mask_y)", "wgrad group C=64 48x36"), ("dgrad C=64 48x36 (mask_y + addend + mask_z)", "wgrad group C=32 96x72"),
             ("dgrad C=64 48x36 (mask_y)", "dgrad C=128 24x18 (mask_y)"), ("forward C=64 48x36", "forward C=32 96x72"),
             ("forward C=64 48x36", "forward C=128 24x18"), ("dgrad C=128 24x18 (mask_y)", "wgrad group C=32 96x72"),
             ("forward C=64 48x36", "forward C=64 48x36"), ("dgrad C=32 96x72 (mask_y)", "wgrad group C=64 48x36"),
             ("dgrad C=32 96x72 (mask_y)", "wgrad group C=32 96x72"), ("forward C=256 12x9", "forward C=32 96x72")]
    for a, b in pairs:
        ta, tb = time_pair(ops[a], ops[b], sa, sb)
        print(f"pair   {a:46s} {ta:7.2f} us (x{ta / alone[a]:.2f})  ||  {b:26s} {tb:7.2f} us (x{tb / alone[b]:.2f})   sum alone {alone[a] + alone[b]:6.1f}, pair wall/launch {max(ta, tb):6.1f}")


if __name__ == "__main__":
    main()
