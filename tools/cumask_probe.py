"""GPU box: which compute units does bit i of a hipExtStreamCreateWithCUMask mask name?  For a few masks the probe
kernel (stl_probe_placement: 4096 blocks that spin ~20 us and report XCC_ID / HW_ID) is run on a masked stream; the
distinct (xcc, se, sh, cu) tuples per mask are printed.  Usage: python tools/cumask_probe.py"""
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stlpose_amd import capi  # noqa: E402


def run(mask_bits, nblocks=4096):
    words = (C.c_uint32 * 8)()
    for i in mask_bits:
        words[i >> 5] |= 1 << (i & 31)
    h = C.c_void_p()
    capi.call("stl_stream_create_masked", words, 8, C.byref(h))
    out = torch.zeros(2 * nblocks, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    capi.call("stl_probe_placement", out.data_ptr(), nblocks, 2000, h.value)
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32").reshape(-1, 2)
    places = collections.Counter()
    for xcc, hw in o:
        places[(int(xcc), (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 15)] += 1
    capi.call("stl_stream_destroy", h.value)
    return places


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    for name, bits in (("all 256", range(256)), ("bits 0..7", range(8)), ("bits 0..31", range(32)), ("bits 0..63", range(64)),
                       ("bits 64..255", range(64, 256)), ("bits 0,8,16,..", range(0, 256, 8)), ("bits 128..255", range(128, 256))):
        pl = run(list(bits))
        per_xcc = collections.Counter(k[0] for k in pl)
        print(f"{name}: {len(pl)} distinct CUs; per XCC {dict(sorted(per_xcc.items()))}")
        if len(pl) <= 32:
            print("   ", sorted(pl))
