"""Build libstlpose_hip.so (gfx950) in-tree with hipcc.  `python -m stlpose_amd.build`."""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "libstlpose_hip.so")
# (source, object tag, extra flags): conv_core and wgrad are compiled once per dtype (-DSTL_DT = number of the TRANSLATION UNIT:
# 0 fp32, 1 bf16 + C ABI, 3 f16 -- not an element-type code, STL_F16 is 2) so that their kernel
# instantiations build in parallel
UNITS = [("capi.hip", "capi", []), ("conv_core.hip", "conv_core_bf16", ["-DSTL_DT=1"]), ("conv_core.hip", "conv_core_f32", ["-DSTL_DT=0"]),
         ("conv_core.hip", "conv_core_f16", ["-DSTL_DT=3"]),
         ("wgrad.hip", "wgrad_bf16", ["-DSTL_DT=1"]), ("wgrad.hip", "wgrad_f32", ["-DSTL_DT=0"]), ("elementwise.hip", "elementwise", []),
         ("program.hip", "program", [])]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value"]


def source_files():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.cuh"))
                  + glob.glob(os.path.join(INCLUDE, "*.h")))


def source_id() -> str:
    """16 hex digits over names and contents of csrc/* and include/*: what stl_build_id() of a current library returns."""
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _digest(deps, flags) -> str:
    """Content hash of an object's inputs (sources it includes + its flags).  Rebuild decisions are by CONTENT, like the build
    id: a checkout or an `rsync -t` that changes a source while keeping an older mtime must still rebuild."""
    h = hashlib.sha256(" ".join(flags).encode())
    for f in deps:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _load_state(path: str) -> dict:
    try:
        import json
        return json.load(open(path))
    except Exception:
        return {}


def build(force: bool = False, verbose: bool = True, stamps: bool = False) -> str:
    """stamps=True: a second library, libstlpose_hip_stamps.so, with the in-kernel phase stamps compiled in
    (-DSTL_STAMPS; tools/conv_stamps.py and tools/wgrad_probe.py load it through STLPOSE_HIP_LIB).  The product
    library never carries them: see the note at STAMP() in conv_core.hip."""
    lib = LIB.replace(".so", "_stamps.so") if stamps else LIB
    suffix, extra = (".stamps.o", ["-DSTL_STAMPS"]) if stamps else (".o", [])
    srcs = source_files()
    sid = source_id()
    state_path = os.path.join(CSRC, ".build_state.json")   # object / library name -> digest of what it was built from
    state = _load_state(state_path)
    objs, jobs, digests = [], [], {}
    for s, tag, fl in UNITS:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, tag + suffix)
        objs.append(obj)
        # capi carries the build id, i.e. depends on every source; the others on their file and the shared headers / includes
        deps = srcs if tag == "capi" else [src] + [f for f in srcs if not f.endswith(".hip")]
        if tag == "capi":
            fl = fl + [f'-DSTL_BUILD_ID="{sid}"']
        cmd = [HIPCC, *FLAGS, *extra, *fl, "-c", src, "-o", obj]
        digests[os.path.basename(obj)] = _digest(deps, cmd[:-1])
        if force or not os.path.exists(obj) or state.get(os.path.basename(obj)) != digests[os.path.basename(obj)]:
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    lib_digest = hashlib.sha256(" ".join(digests[os.path.basename(o)] for o in objs).encode()).hexdigest()[:16]
    if force or jobs or not os.path.exists(lib) or state.get(os.path.basename(lib)) != lib_digest:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    state.update(digests)
    state[os.path.basename(lib)] = lib_digest
    import json
    with open(state_path, "w") as f:
        json.dump(state, f, indent=0, sort_keys=True)
    if verbose:
        print("build id", sid)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
