"""Builds libdspmap_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off keeps a*b+c as
two roundings so that geometry/index math matches the strict CPU oracle bit for
bit ("numerics").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdspmap_hip.so")
SOURCES = ["dspmap_kernels.hip", "dspmap_sweep.hip", "dspmap_api.hip", "dspmap_mgpu.hip", "dspmap_preprocess.hip", "dspmap_velest.hip", "dspmap_dist.hip",
           "velocity_estimator.cpp"]
HEADERS = ["dspmap_internal.h", "dspmap_types.h", "dspmap_device.h", "dspmap_kernels.h", "dspmap_birth.h", "velocity_estimator.h",
           os.path.join("..", "..", "include", "dspmap.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [hipcc] + FLAGS + os.environ.get("DSPMAP_EXTRA_FLAGS", "").split() + ["-x", "hip"] + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
