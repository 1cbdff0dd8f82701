import os, sys
os.environ["STL_CONV_STAMPS"] = "1"
_st = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stlpose_amd", "libstlpose_hip_stamps.so")
assert os.path.exists(_st), "build the stamped library first: python -m stlpose_amd.build --stamps"
os.environ.setdefault("STLPOSE_HIP_LIB", _st)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from stlpose_amd import capi
from tools.conv_probe import run
names = ["consts", "descr", "wres-setup", "tile0 setup+issue+sync", "write_lds+sync", "next setup+issue", "mfma+sync", "(gap)", "epilogue", "loop exit", "stats flush"]
shapes = [(32, 96, 72, 32, 32, 3, 1), (32, 48, 36, 64, 64, 3, 1)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for shape in shapes:
    run(*shape, mode=os.environ.get('MODE', 'bn'), reps=int(os.environ.get('REPS','2')))
    torch.cuda.synchronize()
    buf = (C.c_longlong * 14)()
    capi.call("stl_debug_conv_stamps", C.cast(buf, C.c_void_p))
    t = list(buf)
    print(f"   shader clock = {(t[13]-t[12])/((t[11]-t[0])/100)/1e3:.2f} GHz over {(t[11]-t[0])/100:.1f} us")
    print("   phases (us):", ", ".join(f"{n}={(t[i+1]-t[i])/100:.2f}" for i, n in enumerate(names)), f" total={(t[11]-t[0])/100:.2f}")
