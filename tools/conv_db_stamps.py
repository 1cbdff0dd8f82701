"""Per-stage phase stamps of the two-image conv block (shape 3): block 0 / thread 0 of one launch.
needs the stamped library: python -m stlpose_amd.build --stamps
usage: [SHAPES="32,48,36,64,64,3,1;..."] [MODE=bn|dgradA|dgradB] python tools/conv_db_stamps.py"""
import os, sys
os.environ["STL_CONV_STAMPS"] = "1"
_st = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stlpose_amd", "libstlpose_hip_stamps.so")
assert os.path.exists(_st), "build the stamped library first: python -m stlpose_amd.build --stamps"
os.environ.setdefault("STLPOSE_HIP_LIB", _st)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from stlpose_amd import capi
from tools.conv_probe import run
shapes = [(32, 48, 36, 64, 64, 3, 1), (32, 24, 18, 128, 128, 3, 1), (32, 96, 72, 64, 64, 3, 1)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for mode in os.environ.get("MODE", "bn,dgradA").split(","):
    for shape in shapes:
        z = (C.c_longlong * 64)()
        run(*shape, mode=mode, reps=int(os.environ.get("REPS", "2")))
        torch.cuda.synchronize()
        b1, b2 = (C.c_longlong * 14)(), (C.c_longlong * 64)()
        capi.call("stl_debug_conv_stamps", C.cast(b1, C.c_void_p))
        capi.call("stl_debug_conv_stamps2", C.cast(b2, C.c_void_p))
        t, u = list(b1), list(b2)
        print(f"{mode} {shape}: kernel {(t[11]-t[0])/100:.2f} us; prologue: consts={(t[1]-t[0])/100:.2f} descr={(t[2]-t[1])/100:.2f} "
              f"to-loop={(u[0]-t[2])/100:.2f}; flush={(t[11]-t[10])/100:.2f}")
        i = 0
        while 4 * i + 3 < 64 and u[4 * i + 3] > u[4 * i] > 0 and (i == 0 or u[4 * i] >= u[4 * i - 1]):
            print(f"    stage {i}: taps+slots={(u[4*i+1]-u[4*i])/100:.2f} barrier={(u[4*i+2]-u[4*i+1])/100:.2f} epilogue={(u[4*i+3]-u[4*i+2])/100:.2f}")
            i += 1
