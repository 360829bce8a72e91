"""Build libdcreg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# DCREG_BUILD_TAG=<tag> (with DCREG_EXTRA_FLAGS=-D...) builds an experiment variant next to the product library:
# lib/libdcreg_hip_<tag>.so, loaded when DCREG_LIB points at it (dcreg_amd/api.py).  The product is the untagged library.
_TAG = os.environ.get("DCREG_BUILD_TAG", "")
LIB = os.path.join(LIBDIR, "libdcreg_hip%s.so" % (("_" + _TAG) if _TAG else ""))
OBJDIR = os.path.join(HERE, "build" + (("_" + _TAG) if _TAG else ""))
BINDIR = os.path.join(HERE, "bin")
RUNNER = os.path.join(BINDIR, "icp_test_runner")

SOURCES = [
    "device/context.hip",
    "device/metrics.hip",
    "device/kdtree.hip",
    "device/exchange.hip",
    "host/solver.cpp",
    "host/engine.cpp",
]
HEADERS = ["device/kernels.hpp", "device/search.hpp", "device/context.hpp", "host/linalg.hpp", "host/se3.hpp",
           "../../include/dcreg_debug.h", "../../include/dcreg.h"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(a, deps):
    if not os.path.exists(a):
        return True
    t = os.path.getmtime(a)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", src, "-o", obj,
                   "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("DCREG_EXTRA_FLAGS", "").split()
            if s.endswith(".cpp"):
                cmd[1:1] = ["-x", "c++", "-fopenmp"]   # pure host translation units (OpenMP: per-trial host steps)
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fopenmp", "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if _TAG:
        return LIB
    # the experiment driver (icp_test_runner surface) on top of the C-ABI
    os.makedirs(BINDIR, exist_ok=True)
    rsrc = os.path.join(CSRC, "runner", "test_runner.cpp")
    rdeps = [rsrc, os.path.join(CSRC, "runner", "yaml_lite.hpp"), os.path.join(CSRC, "runner", "pcd_io.hpp"), hdrs[-1], LIB]
    if force or _newer(RUNNER, rdeps):
        cmd = [hipcc, "-x", "c++", "-O2", "-std=c++17", rsrc, "-x", "none", "-o", RUNNER, "-L" + LIBDIR, "-ldcreg_hip",
               "-Wl,-rpath,$ORIGIN/../lib", "-Wall"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
