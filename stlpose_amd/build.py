"""Build libstlpose_hip.so (gfx950) in-tree with hipcc.  `python -m stlpose_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstlpose_hip.so")
SOURCES = ["capi.hip", "conv_core.hip", "wgrad.hip", "elementwise.hip", "program.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, stamps: bool = False) -> str:
    """stamps=True: a second library, libstlpose_hip_stamps.so, with the in-kernel phase stamps compiled in
    (-DSTL_STAMPS; tools/conv_stamps.py and tools/wgrad_probe.py load it through STLPOSE_HIP_LIB).  The product
    library never carries them: see the note at STAMP() in conv_core.hip."""
    lib = LIB.replace(".so", "_stamps.so") if stamps else LIB
    suffix, extra = (".stamps.o", ["-DSTL_STAMPS"]) if stamps else (".o", [])
    hdrs = [os.path.join(CSRC, "common.cuh"), os.path.join(CSRC, "conv_common.inc"), os.path.join(CSRC, "conv_ws.inc"), os.path.join(CSRC, "conv1x1.inc"),
            os.path.join(HERE, "..", "include", "stlpose_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", suffix))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
