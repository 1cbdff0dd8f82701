"""Micro-benchmark: backward of one 3x3 stride-1 C->C conv -- stand-alone data gradient + weight gradient vs the fused launch."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import capi
from stlpose_amd.engine import choose_tile

def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def run(B, H, W, Cc, nsplit, plain_h=True):
    td, dev = torch.bfloat16, "cuda"
    n = B * H * W * Cc
    x = torch.randn(n, device=dev).to(td); dt = torch.randn(n, device=dev).to(td); y = torch.randn(n, device=dev).to(td)
    wb = (torch.randn(Cc * 9 * Cc, device=dev) / math.sqrt(9 * Cc)).to(td)
    st = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device=dev); st[Cc:2 * Cc] = B * H * W
    rst = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device=dev)
    red = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device=dev)
    ga, be = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    out = torch.empty(n, device=dev, dtype=td)
    s = torch.cuda.current_stream().cuda_stream
    def bn(t):
        r = capi.Src(); r.x, r.mode, r.relu = t.data_ptr(), capi.SRC_BN, 1
        r.stats, r.gamma, r.beta, r.inv_count, r.eps = st.data_ptr(), ga.data_ptr(), be.data_ptr(), 1.0 / (B * H * W), 1e-5
        return r
    gs = capi.Src(); gs.x, gs.y, gs.mode = dt.data_ptr(), y.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma, gs.inv_count, gs.eps = st.data_ptr(), rst.data_ptr(), ga.data_ptr(), 1.0 / (B * H * W), 1e-5
    hs = capi.Src()
    if plain_h: hs.x, hs.mode = x.data_ptr(), capi.SRC_PLAIN
    else: hs = bn(x)
    def mk(fused):
        d = capi.Conv(); d.shape = -1
        d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = capi.BF16, B, H, W, Cc, H, W, Cc
        d.ks, d.stride, d.stuff, d.TH, d.TW = 3, 1, 0, 0, 0
        if fused: d.partial = 1
        capi.call("stl_conv_plan", C.byref(d))
        d.src, d.w, d.out = gs, wb.data_ptr(), out.data_ptr()
        d.mask_y, d.mask_bn, d.red = x.data_ptr(), bn(x), red.data_ptr()
        return d
    d0 = mk(False)
    wg = capi.Wgrad()
    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co, wg.ks, wg.stride = 1, B, H, W, Cc, H, W, Cc, 3, 1
    wg.TH, wg.TW = choose_tile(B, H, W, 1, 3, 2, bn_cols=32, maxhalo=576)
    npt = math.ceil(B * (H + 1) / wg.TH) * math.ceil(W / wg.TW)
    chunks = (Cc // 32) ** 2
    wg.nsplit = max(1, min(npt, 256 // chunks))
    part = torch.empty(max(wg.nsplit, nsplit) * Cc * 9 * Cc, device=dev)
    wg.h, wg.g, wg.partial = hs, gs, part.data_ptr()
    t_d = timeit(lambda: capi.call("stl_conv_forward", C.byref(d0), s))
    t_w = timeit(lambda: capi.call("stl_conv_wgrad", C.byref(wg), s))
    d1 = mk(True); d1.partial, d1.wg_nsplit, d1.wg_h = part.data_ptr(), nsplit, hs
    t_f = timeit(lambda: capi.call("stl_conv_forward", C.byref(d1), s))
    msg = f"B{B} {H}x{W} C{Cc} {'plain' if plain_h else 'bn'}-h: dgrad {t_d:6.1f} us (tile {d0.TH}x{d0.TW} shape {d0.shape})  wgrad {t_w:6.1f} us  sum {t_d + t_w:6.1f}  FUSED {t_f:6.1f} us (tile {d1.TH}x{d1.TW}, nsplit {nsplit})"
    if os.environ.get("STL_CONV_STAMPS"):
        capi.call("stl_conv_forward", C.byref(d1), s); torch.cuda.synchronize()
        buf = (C.c_longlong * 14)(); capi.call("stl_debug_conv_stamps", C.cast(buf, C.c_void_p)); t = list(buf)
        names = ["consts", "descr", "wres-setup", "tile0 issue+sync", "write_lds+sync", "next issue", "mfma+sync", "(gap)", "epilogue", "loop exit", "flush+slab"]
        msg += "\n    stamps(us): " + ", ".join(f"{nm}={(t[i+1]-t[i])/100:.2f}" for i, nm in enumerate(names)) + f" total={(t[11]-t[0])/100:.2f}"
    print(msg, flush=True)

if __name__ == "__main__":
    for ns in (256, 512):
        run(32, 96, 72, 32, ns)
        run(32, 96, 72, 32, ns, plain_h=False)
        run(32, 48, 36, 64, ns // 2)
