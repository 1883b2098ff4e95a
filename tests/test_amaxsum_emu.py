"""Asynchronous Max-Sum on the emulated engine build (the very same amaxsum.hip, g++ against the
fake HIP runtime and serial stand-ins for the two hipCUB primitives, tests/emu/hipcub/) against the
oracle, bit for bit -- the CPU twin of tests/test_gpu_amaxsum.py, on smaller instances."""
import os

import pytest

from amaxsum_common import amaxsum_cases, check_golden, compare_amaxsum, golden_files
from pydcop_amd.amaxsum import AMaxSumEngine
from pydcop_amd.graph import Params


def _cases():
    cs = amaxsum_cases(k=3)
    return [(c, "f64") for c in cs] + [(c, "f32") for c in cs[:2] + cs[4:5]]


@pytest.mark.parametrize("case,dtype", _cases(), ids=lambda x: x if isinstance(x, str) else x[0])
def test_amaxsum_emu_bit_exact_vs_oracle(case, dtype, oracle_built):
    from emu.build_emu import build
    from oracle.amaxsum_oracle import OracleAMaxSum
    name, make, kw = case
    g = make()
    p = Params(dtype=dtype, **kw)
    compare_amaxsum(AMaxSumEngine(g, p, lib_path=build()), OracleAMaxSum(g, p),
                    first=(1, 2, 3, 6), last_generation=12, largest=1500)


@pytest.mark.parametrize("case", amaxsum_cases(k=2)[:6], ids=lambda c: c[0])
def test_amaxsum_emu_per_message_handler(case, oracle_built, monkeypatch):
    """MAXSUM_AMAXSUM_GENERIC=1: every destination on the per-message handler (what large domains,
    n-ary factors and degrees above 64 run on)."""
    from emu.build_emu import build
    from oracle.amaxsum_oracle import OracleAMaxSum
    monkeypatch.setenv("MAXSUM_AMAXSUM_GENERIC", "1")
    name, make, kw = case
    g = make()
    p = Params(**kw)
    compare_amaxsum(AMaxSumEngine(g, p, lib_path=build()), OracleAMaxSum(g, p),
                    first=(1, 2, 3, 6), last_generation=24, largest=20_000)


@pytest.mark.parametrize("path", [f for f in golden_files() if not f.endswith("syn_coloring60_Gend.npz")],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_amaxsum_emu_golden_reference_vectors(path):
    from emu.build_emu import build
    check_golden(lambda g, p: AMaxSumEngine(g, p, lib_path=build()), path)


@pytest.mark.parametrize("env", [{"MAXSUM_AMAXSUM_CLEAR_SLOTS": "1"}, {"MAXSUM_AMAXSUM_TWO_SCANS": "1"},
                                 {"MAXSUM_AMAXSUM_ORDER": "dynamic"}, {"MAXSUM_AMAXSUM_ORDER": "static"}, {"MAXSUM_AMAXSUM_CAP_LOOKUP": "1"},
                                 {"MAXSUM_AMAXSUM_CLEAR_SLOTS": "1", "MAXSUM_AMAXSUM_GENERIC": "1"}],
                         ids=lambda e: "+".join(k[15:].lower() + "=" + v for k, v in e.items()))
def test_amaxsum_bookkeeping_variants(env, oracle_built, monkeypatch):
    """The switches of the generation bookkeeping (amaxsum.hip, step / finish): slot words poisoned before the
    handlers run (every handler must write the words of all its output slots -- they are not cleared), the two
    separate scans of very large generations, the destinations re-ordered by queue length / in the static order."""
    from emu.build_emu import build
    from oracle.amaxsum_oracle import OracleAMaxSum
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, make, kw in amaxsum_cases(k=2)[:5:2]:  # (three cases here; the GPU twin runs five)
        g = make()
        p = Params(**kw)
        compare_amaxsum(AMaxSumEngine(g, p, lib_path=build()), OracleAMaxSum(g, p), first=(1, 2, 3, 6), last_generation=10, largest=5_000)
