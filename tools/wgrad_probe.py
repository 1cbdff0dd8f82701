"""Weight-gradient kernel timing vs split factor: separates per-tile from fixed per-launch cost."""
import os, sys, math, ctypes as C
_st = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stlpose_amd", "libstlpose_hip_stamps.so")
if os.environ.get("STL_CONV_STAMPS") and os.path.exists(_st):   # stamps exist only in the -DSTL_STAMPS build
    os.environ.setdefault("STLPOSE_HIP_LIB", _st)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import capi
from stlpose_amd.engine import choose_tile

def run(B, H, W, Ci, Co, ks, maxpx, splits, gq=True):
    dev = "cuda"
    f32 = os.environ.get("DTYPE", "bf16") == "fp32"
    td = torch.float32 if f32 else torch.bfloat16
    x = torch.randn(B * H * W * Ci, device=dev).to(td)
    dt = torch.randn(B * H * W * Co, device=dev).to(td)
    y = torch.randn(B * H * W * Co, device=dev).to(td)
    st = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device=dev); st[Co:2 * Co] = B * H * W
    st1 = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device=dev); st1[Ci:2 * Ci] = B * H * W
    rst = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device=dev)
    ga = torch.ones(max(Ci, Co), device=dev); be = torch.zeros(max(Ci, Co), device=dev)
    wg = capi.Wgrad()
    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = (0 if f32 else 1), B, H, W, Ci, H, W, Co
    wg.ks, wg.stride = ks, 1
    wg.TH, wg.TW = choose_tile(B, H, W, 1, ks, 4 if f32 else 2, bn_cols=32, maxpx=maxpx, maxhalo=384 if maxpx == 256 else 576)
    npt = math.ceil(B * (H + 1) / wg.TH) * math.ceil(W / wg.TW)
    wg.h.x, wg.h.mode, wg.h.relu = x.data_ptr(), capi.SRC_BN, 1
    wg.h.stats, wg.h.gamma, wg.h.beta, wg.h.inv_count, wg.h.eps = st1.data_ptr(), ga.data_ptr(), be.data_ptr(), 1.0 / (B * H * W), 1e-5
    wg.g.x, wg.g.y, wg.g.mode = dt.data_ptr(), y.data_ptr(), capi.SRC_BNBWD if gq else capi.SRC_PLAIN
    wg.g.stats, wg.g.rstats, wg.g.gamma, wg.g.inv_count, wg.g.eps = st.data_ptr(), rst.data_ptr(), ga.data_ptr(), 1.0 / (B * H * W), 1e-5
    chunks = math.ceil(Co / 32) * math.ceil(Ci / 32)
    part = torch.empty(max(splits) * Co * ks * ks * Ci, device=dev)
    wg.partial = part.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    for ns in splits:
        if ns > npt: continue
        wg.nsplit = ns
        for _ in range(5): capi.call("stl_conv_wgrad", C.byref(wg), s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): capi.call("stl_conv_wgrad", C.byref(wg), s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        st = ""
        if os.environ.get("STL_CONV_STAMPS"):
            buf = (C.c_longlong * 16)()
            capi.call("stl_debug_wgrad_stamps", C.cast(buf, C.c_void_p))
            v = list(buf)
            st = " stamps(us): " + " ".join(f"{(v[i] - v[0]) / 100.0:.1f}" for i in range(1, 9))
        print(f"{B}x{H}x{W} Ci{Ci} Co{Co} k{ks} tile {wg.TH}x{wg.TW} npt {npt} nsplit {ns:4d} blocks {ns*chunks:4d} tiles/blk {math.ceil(npt/ns):3d}  {us:7.1f} us{st}", flush=True)

if __name__ == "__main__" and len(sys.argv) > 1:
    a = [int(v) for v in sys.argv[1].split(",")]
    run(a[0], a[1], a[2], a[3], a[4], a[5], a[7], [a[6]])
    sys.exit(0)
for maxpx in ((128, 256) if __name__ == "__main__" else ()):
    run(32, 96, 72, 32, 32, 3, maxpx, [64, 128, 256, 384, 512])
    run(32, 48, 36, 64, 64, 3, maxpx, [16, 32, 64, 96, 128])
    run(32, 24, 18, 128, 128, 3, maxpx, [4, 8, 16, 24, 32])
    run(32, 12, 9, 256, 256, 3, maxpx, [2, 4, 6, 8])
