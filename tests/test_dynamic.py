"""pydcop_amd.dynamic (maxsum_dynamic.py restated on the engine) on the CPU: the scope-change
bookkeeping against a dict-based restatement of the reference's handlers, and a run with every
kind of change through the emulated engine, bit-exact against the oracle."""
import pytest

from dynamic_common import check_dynamic_run, check_rescope_semantics


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_rescope_bookkeeping_follows_the_reference_handlers(seed):
    check_rescope_semantics(seed)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("float_tables", [True, False])
def test_dynamic_run_on_the_emulated_engine(dtype, float_tables, oracle_built):
    from emu.build_emu import build
    check_dynamic_run(oracle_built, lib_path=build(), dtype=dtype, float_tables=float_tables)
