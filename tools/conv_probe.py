"""Micro-benchmark of one conv launch (dominant HRNet shapes) with tuning knobs from the environment."""
import ctypes as C, math, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stlpose_amd import capi
from stlpose_amd.engine import choose_tile

def run(B, H, W, Ci, Co, ks, s, mode="bn", reps=30, tile=None):
    td = torch.bfloat16
    stuff = mode.startswith("s2")     # s2dA / s2dB: data gradient of a stride-2 conv, H x W is its OUTPUT (the forward conv's input)
    if stuff:
        mode = "dgrad" + mode[3:]
        Hs, Ws = (H + 1) // 2, (W + 1) // 2
    x = torch.randn(B, (Hs if stuff else H), (Ws if stuff else W), Ci, device="cuda").to(td)
    w = (torch.randn(Co, ks * ks, Ci, device="cuda") / math.sqrt(Ci * ks * ks)).to(td)
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // s + 1, (W + 2 * pad - ks) // s + 1
    if stuff:
        Ho, Wo = H, W
        H, W = Hs, Ws
    out = torch.empty(B, Ho, Wo, Co, device="cuda", dtype=td)
    st = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    sx = torch.zeros(capi.NSHARD, 2, Ci, dtype=torch.float64, device="cuda")
    xf = x.float().reshape(-1, Ci).double()
    sx[0, 0], sx[0, 1] = xf.sum(0), (xf * xf).sum(0)
    ga, be = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = capi.BF16, B, H, W, Ci, Ho, Wo, Co
    p.ks, p.stride = ks, s
    p.stuff = int(stuff)
    p.TH, p.TW = tile or (0, 0)
    p.src.x = x.data_ptr()
    if mode == "bn":
        p.src.mode, p.src.relu = capi.SRC_BN, 1
        p.src.stats, p.src.gamma, p.src.beta = sx.data_ptr(), ga.data_ptr(), be.data_ptr()
        p.src.inv_count, p.src.eps = 1.0 / (B * H * W), 1e-5
    p.w, p.out, p.out_stats = w.data_ptr(), out.data_ptr(), st.data_ptr()
    keep = []
    if mode in ("dgradA", "dgradB"):   # data gradient of a two-conv unit: BNBWD source, mask (+ addend, block-end mask) epilogue
        yq = torch.randn(B, H, W, Ci, device="cuda").to(td)
        rst = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
        p.src.mode, p.src.relu = capi.SRC_BNBWD, 0
        p.src.y, p.src.stats, p.src.rstats, p.src.gamma = yq.data_ptr(), sx.data_ptr(), rst.data_ptr(), ga.data_ptr()
        p.src.inv_count, p.src.eps = 1.0 / (B * H * W), 1e-5
        my = torch.randn(B, Ho, Wo, Co, device="cuda").to(td)
        so = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda"); so[0, 1] = B * Ho * Wo
        go, bo = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
        red = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
        p.out_stats = None
        p.mask_y = my.data_ptr()
        p.mask_bn.x, p.mask_bn.mode, p.mask_bn.relu = my.data_ptr(), capi.SRC_BN, int(mode == "dgradA")
        p.mask_bn.stats, p.mask_bn.gamma, p.mask_bn.beta = so.data_ptr(), go.data_ptr(), bo.data_ptr()
        p.mask_bn.inv_count, p.mask_bn.eps = 1.0 / (B * Ho * Wo), 1e-5
        p.red = red.data_ptr()
        keep += [yq, rst, my, so, go, bo, red]
        if mode == "dgradB":
            ad = torch.randn(B, Ho, Wo, Co, device="cuda").to(td)
            mz = torch.randn(B, Ho, Wo, Co, device="cuda").to(td)
            p.addend, p.mask_z = ad.data_ptr(), mz.data_ptr()
            keep += [ad, mz]
    capi.call("stl_conv_plan", C.byref(p))
    capi.call("stl_conv_plan", C.byref(p))
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        capi.call("stl_conv_forward", C.byref(p), stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        capi.call("stl_conv_forward", C.byref(p), stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * B * Ho * Wo * Co * Ci * ks * ks
    by = B * H * W * Ci * 2 + B * Ho * Wo * Co * 2
    print(f"{mode:6s} B{B} {H}x{W} {Ci}->{Co} k{ks}s{s} tile={p.TH}x{p.TW} cap={os.environ.get('STL_CONV_GRID_CAP','-')}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e3:7.1f} GB/s", flush=True)

if __name__ == "__main__":
    shapes = [(32, 96, 72, 32, 32, 3, 1), (32, 48, 36, 64, 64, 3, 1), (32, 24, 18, 128, 128, 3, 1), (32, 12, 9, 256, 256, 3, 1),
              (32, 96, 72, 64, 256, 1, 1), (32, 96, 72, 256, 64, 1, 1)]
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        for sh in shapes:
            run(*sh)
    else:
        for cap in ("256", "512", "768", "1024", "1280", "2560"):
            env = dict(os.environ, STL_CONV_GRID_CAP=cap)
            subprocess.run([sys.executable, __file__, "one"], env=env)
