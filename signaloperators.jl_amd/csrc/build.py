"""Builds libsigops.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = ["k_pointwise.hip", "k_sos.hip", "k_resample.hip", "kernels2.hip", "planner.cpp", "stages.cpp", "accumulator.cpp", "executor.cpp", "design.cpp", "capi.cpp", "comm.cpp"]
HDRS = ["kernels.h", "kcommon.h", "plan.h", "plan_impl.h", "sigops_internal.h", "../../include/sigops.h"]  # (flag changes in this file: --force)
OUT = os.path.join(HERE, "libsigops.so")


def _mtime(rel):
    return os.path.getmtime(os.path.join(HERE, rel))


def _obj(src):
    return os.path.join(HERE, src.rsplit(".", 1)[0] + ".o")


def _stale(src):
    """an object is rebuilt when its source or a header it includes is newer (k_resample.hip takes
    minutes: it must not be recompiled for a change to the planner's headers)"""
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    hdrs = [h for h in HDRS if not (src.endswith(".hip") and h in ("plan.h", "plan_impl.h"))]
    return _mtime(src) > t or any(_mtime(h) > t for h in hdrs)


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_mtime(d) > t for d in SRCS + HDRS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(s):
        o = _obj(s)
        if not force and not _stale(s):
            return o
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
               "-Wno-unused-function", "-x", "hip", "-c", os.path.join(HERE, s), "-o", o]
        if s.endswith(".hip"):
            # MachineLICM hoists the fp64 polynomial constants of sin/cos out of the resampler's
            # loader loop into VGPR pairs and then SPILLS them; every scratch reload waits
            # vmcnt(0) and drains the LDS-DMA ring (measured: tile issue took a full transfer)
            cmd[1:1] = ["-mllvm", "-disable-machine-licm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    # the translation units are independent: compile them side by side (k_resample.hip dominates)
    with ThreadPoolExecutor(max_workers=len(SRCS)) as pool:
        objs = list(pool.map(compile_one, SRCS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
