"""What does running two launches side by side buy?  Pairs / quadruples of independent conv and weight-gradient
launches on separate streams vs the same launches back to back on one stream (HIP events, 50 repetitions)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import capi
from stlpose_amd.engine import choose_tile

KEEP = []


def conv_inst(B, H, W, Ci, Co, ks=3, s=1):
    td = torch.bfloat16
    x = torch.randn(B, H, W, Ci, device="cuda").to(td)
    w = (torch.randn(Co, ks * ks, Ci, device="cuda") / math.sqrt(Ci * ks * ks)).to(td)
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // s + 1, (W + 2 * pad - ks) // s + 1
    out = torch.empty(B, Ho, Wo, Co, device="cuda", dtype=td)
    st = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    sx = torch.zeros(capi.NSHARD, 2, Ci, dtype=torch.float64, device="cuda")
    sx[0, 1] = B * H * W
    ga, be = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = capi.BF16, B, H, W, Ci, Ho, Wo, Co
    p.ks, p.stride = ks, s
    p.src.x, p.src.mode, p.src.relu = x.data_ptr(), capi.SRC_BN, 1
    p.src.stats, p.src.gamma, p.src.beta = sx.data_ptr(), ga.data_ptr(), be.data_ptr()
    p.src.inv_count, p.src.eps = 1.0 / (B * H * W), 1e-5
    p.w, p.out, p.out_stats = w.data_ptr(), out.data_ptr(), st.data_ptr()
    capi.call("stl_conv_plan", C.byref(p))
    KEEP.append((x, w, out, st, sx, ga, be, p))
    return ("stl_conv_forward", p, f"conv{Ci}@{H}x{W}")


def wgrad_inst(B, H, W, Ci, Co, ks=3, blocks=256):
    dev = "cuda"
    x = torch.randn(B * H * W * Ci, device=dev).bfloat16()
    dt = torch.randn(B * H * W * Co, device=dev).bfloat16()
    y = torch.randn(B * H * W * Co, device=dev).bfloat16()
    st = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device=dev); st[Co:2 * Co] = B * H * W
    st1 = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device=dev); st1[Ci:2 * Ci] = B * H * W
    rst = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device=dev)
    ga = torch.ones(max(Ci, Co), device=dev); be = torch.zeros(max(Ci, Co), device=dev)
    wg = capi.Wgrad()
    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = 1, B, H, W, Ci, H, W, Co
    wg.ks, wg.stride = ks, 1
    wg.TH, wg.TW = choose_tile(B, H, W, 1, ks, 2, bn_cols=32, maxpx=128, maxhalo=576)
    npt = math.ceil(B * (H + 1) / wg.TH) * math.ceil(W / wg.TW)
    wg.h.x, wg.h.mode, wg.h.relu = x.data_ptr(), capi.SRC_BN, 1
    wg.h.stats, wg.h.gamma, wg.h.beta, wg.h.inv_count, wg.h.eps = st1.data_ptr(), ga.data_ptr(), be.data_ptr(), 1.0 / (B * H * W), 1e-5
    wg.g.x, wg.g.y, wg.g.mode = dt.data_ptr(), y.data_ptr(), capi.SRC_BNBWD
    wg.g.stats, wg.g.rstats, wg.g.gamma, wg.g.inv_count, wg.g.eps = st.data_ptr(), rst.data_ptr(), ga.data_ptr(), 1.0 / (B * H * W), 1e-5
    chunks = math.ceil(Co / 32) * math.ceil(Ci / 32)
    wg.nsplit = min(npt, max(1, blocks // chunks))
    part = torch.empty(wg.nsplit * Co * ks * ks * Ci, device=dev)
    wg.partial = part.data_ptr()
    KEEP.append((x, dt, y, st, st1, rst, ga, be, part, wg))
    return ("stl_conv_wgrad", wg, f"wgrad{Ci}@{H}x{W}")


def timeit(insts, streams, reps=50):
    """insts[i] runs on streams[i]; returns us per round (one launch of every instance)"""
    main = torch.cuda.current_stream()
    for _ in range(3):
        for (fn, d, _), s in zip(insts, streams):
            capi.call(fn, C.byref(d), s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for s in set(streams):
        if s is not main:
            s.wait_event(e0)
    for _ in range(reps):
        for (fn, d, _), s in zip(insts, streams):
            capi.call(fn, C.byref(d), s.cuda_stream)
    for s in set(streams):
        if s is not main:
            ev = torch.cuda.Event(); ev.record(s); main.wait_event(ev)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    main = torch.cuda.current_stream()
    extra = [torch.cuda.Stream() for _ in range(3)]
    B = 32
    mk = {
        "c32": lambda: conv_inst(B, 96, 72, 32, 32), "c64": lambda: conv_inst(B, 48, 36, 64, 64),
        "c128": lambda: conv_inst(B, 24, 18, 128, 128), "c256": lambda: conv_inst(B, 12, 9, 256, 256),
        "w32": lambda: wgrad_inst(B, 96, 72, 32, 32), "w64": lambda: wgrad_inst(B, 48, 36, 64, 64),
        "w128": lambda: wgrad_inst(B, 24, 18, 128, 128), "w256": lambda: wgrad_inst(B, 12, 9, 256, 256),
    }
    combos = [("c32", "c32"), ("w32", "w32"), ("c32", "w32"), ("c64", "w32"), ("c32", "c64"), ("c128", "c256"), ("w128", "w256"),
              ("c32", "c64", "c128", "c256"), ("w32", "w64", "w128", "w256"), ("c32", "c64", "w32", "w64"), ("c32", "w32", "w64", "w128")]
    for combo in combos:
        insts = [mk[n]() for n in combo]
        alone = [timeit([i], [main]) for i in insts]
        serial = timeit(insts, [main] * len(insts))
        par = timeit(insts, [main] + extra[:len(insts) - 1])
        print(f"{'+'.join(combo):28s} alone {' '.join(f'{a:5.1f}' for a in alone)}  one stream {serial:6.1f} us  own streams {par:6.1f} us"
              f"  -> overlap x{serial / par:.2f}  (longest alone {max(alone):.1f})", flush=True)
        KEEP.clear()
