"""Builds libraytracer_amd.so (HIP kernels + host code + C ABI) in-tree with hipcc.

The compile needs no GPU: hipcc cross-compiles for gfx950.  Flags that matter
for correctness: -ffp-contract=off (the reference's float32 arithmetic is
never fused and the bit-exact modes depend on it) and
-fhip-fp32-correctly-rounded-divide-sqrt (IEEE 1/x and sqrt on the device).
For speed: -fno-slp-vectorize (left on, the vectoriser pairs up the scalar
arithmetic of the triangle test into packed instructions and spends more moves
on building the register pairs than the packed instructions save).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libraytracer_amd.so")

SOURCES = ["bvh_build.cpp", "bvh_io.cpp", "flatten.cpp", "scene_host.cpp", "obj_loader.cpp",
           "png_decode.cpp", "backend.cpp", "capi.cpp", "rt_kernels.hip"]
HEADERS = ["gomath.h", "json.h", "rt_core.h", "rt_device.h", "rt_internal.h", "backend.h",
           os.path.join("..", "..", "include", "rt_amd.h"),
           os.path.join("..", "..", "include", "rt_backend.h")]

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread",
         "-Wall", "-Wextra"]


def _existing(names):
    return [os.path.join(CSRC, n) for n in names if os.path.exists(os.path.join(CSRC, n))]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _existing(SOURCES) + _existing(HEADERS))


def build_library(force=False, verbose=False):
    """Compile if sources are newer than the library; returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB_PATH] + _existing(SOURCES)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


RCP_CHECK_SRC = os.path.join(os.path.dirname(HERE), "tools", "ubench", "rcp_check.hip")
RCP_CHECK_BIN = os.path.join(os.path.dirname(HERE), "tools", "ubench", "rcp_check")


def build_rcp_check(force=False):
    """The exhaustive check of the triangle test's short reciprocal (rt_core.h:
    recip_det) as a small program of its own; the GPU tests run it."""
    if (not force and os.path.exists(RCP_CHECK_BIN)
            and os.path.getmtime(RCP_CHECK_BIN) >= os.path.getmtime(RCP_CHECK_SRC)):
        return RCP_CHECK_BIN
    hipcc = os.environ.get("HIPCC", "hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result",
                           RCP_CHECK_SRC, "-o", RCP_CHECK_BIN])
    return RCP_CHECK_BIN


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_rcp_check(force="--force" in sys.argv))
