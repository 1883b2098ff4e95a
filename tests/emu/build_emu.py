"""Build tests/emu/_build/libmaxsum_emu.so: the product's engine.hip + layout.cpp
compiled by g++ against the serial fake HIP runtime in tests/emu/hip/ (TEST ONLY)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pydcop_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libmaxsum_emu.so")


def build(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "layout.cpp", "amaxsum.hip", "mgm.hip", "dsa.hip", "bin_box.hip", "small_box.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("kernels.h", "nary_box.h", "bin_box.h", "small_box.h", "layout.h", "local_search.h")] + [
        os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "hipcub", "hipcub.hpp"),
        os.path.join(ROOT, "include", "maxsum_gpu.h")]
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-attributes",
           "-DMXS_EMULATED_HIPCUB", "-DMXS_BUILD_KIND=0",  # (build kind 0: the binding refuses this library outside tests)
           "-I", HERE, "-x", "c++"] + srcs + ["-o", OUT, "-ldl", "-pthread"]
    subprocess.check_call(cmd)
    return OUT


FAKE_RCCL = os.path.join(HERE, "_build", "libfake_rccl.so")


def build_fake_rccl(force=False):
    """tests/emu/fake_rccl: the RCCL entry points the engine binds, over files (TEST ONLY)."""
    src = os.path.join(HERE, "fake_rccl", "fake_rccl.cpp")
    if not force and os.path.exists(FAKE_RCCL) and os.path.getmtime(FAKE_RCCL) >= os.path.getmtime(src):
        return FAKE_RCCL
    os.makedirs(os.path.dirname(FAKE_RCCL), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", FAKE_RCCL])
    return FAKE_RCCL


FAKE_RCCL = os.path.join(HERE, "_build", "libfake_rccl.so")


def build_fake_rccl(force=False):
    """tests/emu/fake_rccl: the RCCL entry points the engine binds, over files (TEST ONLY)."""
    src = os.path.join(HERE, "fake_rccl", "fake_rccl.cpp")
    if not force and os.path.exists(FAKE_RCCL) and os.path.getmtime(FAKE_RCCL) >= os.path.getmtime(src):
        return FAKE_RCCL
    os.makedirs(os.path.dirname(FAKE_RCCL), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", FAKE_RCCL])
    return FAKE_RCCL


if __name__ == "__main__":
    print(build_fake_rccl(force=True))
    print(build_fake_rccl(force=True))
    print(build(force=True))
