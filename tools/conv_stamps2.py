import os, sys
os.environ["STL_CONV_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
from stlpose_amd import capi
from tools.conv_probe import run
for shape in [(32, 96, 72, 64, 64, 3, 1), (32, 96, 72, 32, 32, 3, 1)]:
    run(*shape, reps=2)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    capi.call("stl_debug_conv_stamps2", C.cast(buf, C.c_void_p))
    t = list(buf)
    t0 = t[0]
    for i in range(6):
        l = t[i*4:i*4+4]; c = t[32+i*4:32+i*4+4]
        if l[0] == 0: break
        print(f"  stage {i}: LOADER top@{(l[0]-t0)/100:7.2f} write_lds={(l[1]-l[0])/100:5.2f} setup+issue={(l[2]-l[1])/100:5.2f} barrier_wait={(l[3]-l[2])/100:5.2f} | COMPUTE top@{(c[0]-t0)/100:7.2f} mfma={(c[1]-c[0])/100:5.2f} epi={(c[2]-c[1])/100:5.2f} barrier_wait={(c[3]-c[2])/100:5.2f}")
