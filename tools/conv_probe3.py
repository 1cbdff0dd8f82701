import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, math, torch
from stlpose_amd import capi
def run(B, H, W, Ci, Co, ks, s, mode, stats, reps=30):
    td = torch.bfloat16
    x = torch.randn(B, H, W, Ci, device="cuda").to(td)
    w = (torch.randn(Co, ks * ks, Ci, device="cuda") / math.sqrt(Ci * ks * ks)).to(td)
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // s + 1, (W + 2 * pad - ks) // s + 1
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
    p.src.x = x.data_ptr()
    if mode == "bn":
        p.src.mode, p.src.relu = capi.SRC_BN, 1
        p.src.stats, p.src.gamma, p.src.beta = sx.data_ptr(), ga.data_ptr(), be.data_ptr()
        p.src.inv_count, p.src.eps = 1.0 / (B * H * W), 1e-5
    p.w, p.out = w.data_ptr(), out.data_ptr()
    if stats:
        p.out_stats = st.data_ptr()
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
    print(f"B{B} {H}x{W} {Ci}->{Co} k{ks}s{s} mode={mode} stats={stats}: {us:8.1f} us", flush=True)
for B in (1, 4, 32):
    for mode, stats in (("bn", 1), ("plain", 1), ("plain", 0)):
        run(B, 96, 72, 32, 32, 3, 1, mode, stats)
for B in (1, 32):
    for mode, stats in (("bn", 1), ("plain", 0)):
        run(B, 24, 18, 128, 128, 3, 1, mode, stats)
# empty-ish kernel reference: a torch elementwise op launch cadence
x = torch.zeros(1024, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): x.add_(1.0)
e1.record(); torch.cuda.synchronize()
print("torch tiny add_ cadence us", e0.elapsed_time(e1) * 10)
