"""Per-tile phase stamps of the 3x3 weight-gradient kernel: block 0 / thread 0 of one launch (stamped build:
python -m stlpose_amd.build --stamps).  usage: python tools/wgrad_tile_stamps.py [B,H,W,Ci,Co,nsplit ...]"""
import os, sys, ctypes as C
os.environ["STL_CONV_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.wgrad_probe as w   # selects the stamped library
import torch
from stlpose_amd import capi
cases = [(32, 96, 72, 32, 32, 128), (32, 96, 72, 32, 32, 512), (32, 48, 36, 64, 64, 32), (32, 48, 36, 64, 64, 128), (32, 24, 18, 128, 128, 32)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, H, W, Ci, Co, ns in cases:
    w.run(B, H, W, Ci, Co, 3, 128, [ns])
    torch.cuda.synchronize()
    b = (C.c_longlong * 64)()
    capi.call("stl_debug_wgrad_stamps2", C.cast(b, C.c_void_p))
    u = list(b)
    rows = []
    for i in range(16):
        a = u[4 * i:4 * i + 4]
        if not (a[0] and a[3] > a[0]) or (i and a[0] < u[4 * i - 1]):
            break
        rows.append(f"[write+barrier {(a[1]-a[0])/100:.2f} fetch {(a[2]-a[1])/100:.2f} mfma+barrier {(a[3]-a[2])/100:.2f}]")
    print("   tiles:", " ".join(rows[:8]))
