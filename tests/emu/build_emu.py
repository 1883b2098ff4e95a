"""Build tests/emu/_build/libmaxsum_emu.so: the product's engine.hip + layout.cpp
compiled by g++ against the serial fake HIP runtime in tests/emu/hip/ (TEST ONLY)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pydcop_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libmaxsum_emu.so")


def build(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "layout.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kernels.h", "layout.h")] + [
        os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "maxsum_gpu.h")]
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-attributes",
           "-I", HERE, "-x", "c++", srcs[0], srcs[1], "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
