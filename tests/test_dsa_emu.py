"""DSA on the emulated engine build (the very same dsa.hip, g++ against the fake HIP runtime)
against the oracle, bit for bit -- the CPU twin of tests/test_gpu_dsa.py."""
import pytest

from dsa_common import compare_dsa, dsa_cases
from pydcop_amd.graph import Params


@pytest.mark.parametrize("case", dsa_cases(k=4), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_dsa_emu_bit_exact_vs_oracle(case, dtype, oracle_built):
    from emu.build_emu import build
    from oracle.dsa_oracle import OracleDsa
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(dtype=dtype, **kw), dsa_kw, lib_path=build(), steps=(0, 1, 1, 3, 10))
