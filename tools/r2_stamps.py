"""Phase stamps of block 0 of the two-per-CU conv kernel (conv_r2.inc; stamped library: python -m stlpose_amd.build --stamps).
Per stage: halo wait + transform + LDS write | barrier (filter DMA lands) | MFMAs + barrier.  SHAPES="B,H,W,Ci,Co,ks,s;..." MODE=bn|dgradA|dgradB"""
import os, sys
os.environ["STL_CONV_STAMPS"] = "1"
_st = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stlpose_amd", "libstlpose_hip_stamps.so")
assert os.path.exists(_st), "build the stamped library first: python -m stlpose_amd.build --stamps"
os.environ.setdefault("STLPOSE_HIP_LIB", _st)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from stlpose_amd import capi
from tools.conv_probe import run

shapes = [(32, 48, 36, 64, 64, 3, 1), (32, 24, 18, 128, 128, 3, 1), (32, 12, 9, 256, 256, 3, 1)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for mode in os.environ.get("MODE", "bn,dgradA").split(","):
    for shape in shapes:
        run(*shape, mode=mode, reps=int(os.environ.get("REPS", "2")))
        torch.cuda.synchronize()
        print("   kernel:", capi.lib().stl_last_kernel().decode())
        buf = (C.c_longlong * 64)()
        capi.call("stl_debug_conv_stamps2", C.cast(buf, C.c_void_p))
        t = [v / 100.0 for v in buf]   # us
        ns = (shape[3] + 31) // 32
        us = lambda a, b: t[b] - t[a]   # noqa: E731
        print(f"   start -> first loads issued {us(0, 1):.2f}, constants + barrier {us(1, 2):.2f}")
        prev = 2
        for s_ in range(ns):
            a, b, c = 3 + 3 * s_, 4 + 3 * s_, 5 + 3 * s_
            print(f"   stage {s_}: halo wait + transform + write {us(prev, a):.2f} | barrier (filter DMA lands) {us(a, b):.2f} | MFMAs + barrier {us(b, c):.2f}")
            prev = c
        print(f"   epilogue (fetch, apply, store) {us(prev, 40):.2f}, tile sums {us(40, 41):.2f}, atomics issued {us(41, 42):.2f}; block total {us(0, 42):.2f} us")
