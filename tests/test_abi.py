"""The C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/maxsum_gpu.h declares; and the product fails loudly (no CPU
fallback) when no device is there."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    import __graft_entry__ as ge
    ge.build()
    from pydcop_amd.engine import load_library
    return load_library(os.path.join(ROOT, "pydcop_amd", "csrc", "libmaxsum_hip.so"))


def test_header_symbols_are_exported(hip_lib):
    from pydcop_amd.engine import ABI_SYMBOLS
    header = open(os.path.join(ROOT, "include", "maxsum_gpu.h")).read()
    declared = set(re.findall(r"\b(mxs_[a-z_]+)\s*\(", header))
    assert declared == set(ABI_SYMBOLS)
    for name in declared:
        assert hasattr(hip_lib, name)
    assert hip_lib.mxs_version() >= 100


def test_partition_header_symbols_are_exported(hip_lib):
    """include/maxsum_partition.h <-> libmxs_partition.so (host code, built by build())."""
    import ctypes
    header = open(os.path.join(ROOT, "include", "maxsum_partition.h")).read()
    declared = set(re.findall(r"\b(mxp_[a-z_]+)\s*\(", header))
    assert declared == {"mxp_partition", "mxp_last_error"}
    lib = ctypes.CDLL(os.path.join(ROOT, "pydcop_amd", "csrc", "libmxs_partition.so"))
    for name in declared:
        assert hasattr(lib, name)


def test_no_cpu_fallback_without_gpu(hip_lib):
    from pydcop_amd import generators as G
    from pydcop_amd.engine import MaxSumEngine, MaxSumGpuError, device_count
    from pydcop_amd.graph import Params
    if device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(MaxSumGpuError, match="no CPU fallback|no HIP device"):
        MaxSumEngine(G.random_coloring(10, seed=0), Params(),
                     lib_path=os.path.join(ROOT, "pydcop_amd", "csrc", "libmaxsum_hip.so"))


def test_product_cannot_be_pointed_at_the_emulated_engine(hip_lib):
    """No CPU fallback by parameter or environment variable: outside the tests' own registration
    (pydcop_amd.engine.register_test_engine, a Python call) the host emulation of the engine
    sources is refused -- by name and by what the binary says it is (mxs_build_kind)."""
    import shutil
    import subprocess
    import sys
    import tempfile
    from emu.build_emu import build
    emu = build()
    assert hip_lib.mxs_build_kind() == 1
    with tempfile.TemporaryDirectory() as d:
        disguised = os.path.join(d, "libmaxsum_hip_fast.so")  # an emulated build under a product name
        shutil.copy(emu, disguised)
        for env, arg in (({"MAXSUM_HIP_LIB": emu}, "None"), ({"MAXSUM_HIP_LIB": disguised}, "None"),
                         ({}, repr(emu)), ({}, repr(disguised))):
            code = ("import sys; sys.path.insert(0, %r)\n"
                    "from pydcop_amd.engine import load_library, MaxSumGpuError\n"
                    "try:\n    load_library(%s)\nexcept MaxSumGpuError as e:\n    print('REFUSED', e)\n"
                    "else:\n    print('LOADED')\n" % (ROOT, arg))
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=120)
            assert "REFUSED" in r.stdout and "no CPU fallback" in r.stdout, (env, arg, r.stdout, r.stderr[-500:])


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "pydcop_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                # (comments may cite the checker's source files; nothing may load or call them)
                assert "maxsum_oracle" not in src.replace("oracle/maxsum_oracle.c", "").replace(
                    "oracle/amaxsum_oracle.c", ""), f


def _kernel_scratch(source):
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "pydcop_amd", "csrc")
    out = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Rpass-analysis=kernel-resource-usage", "-c", source, "-o", os.devnull],
        cwd=csrc, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(scratch)
    return list(zip(names, scratch))


def test_no_kernel_uses_scratch():
    """all per-item state lives in registers / LDS.  A kernel that
    indexes a local array dynamically silently gets scratch memory (= extra HBM traffic:
    the n-ary kernel once wrote 644 MB per cycle that way); the compiler's resource
    report must show 0 bytes for every kernel."""
    ks = _kernel_scratch("engine.hip")
    assert len(ks) >= 20
    bad = [(n, s) for n, s in ks if s != 0]
    assert not bad, f"kernels using scratch: {bad}"


@pytest.mark.parametrize("source,at_least", [("dsa.hip", 10), ("mgm.hip", 14), ("amaxsum.hip", 6)])
def test_no_kernel_of_the_other_engines_uses_scratch(source, at_least):
    """The same for the DSA / MGM / A-Max-Sum sources (the register arrays of the slot kernels:
    lsearch::pick).  rocPRIM's own sort kernels (amaxsum.hip) are not ours to judge."""
    ks = [(n, s) for n, s in _kernel_scratch(source) if "rocprim" not in n]
    assert len(ks) >= at_least, ks
    bad = [(n, s) for n, s in ks if s != 0]
    assert not bad, f"kernels using scratch: {bad}"
