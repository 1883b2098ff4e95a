"""Host logic + kernel index arithmetic of the engine, checked on the CPU.

The engine's own sources (pydcop_amd/csrc/engine.hip, kernels.h, layout.cpp) are
compiled by g++ against tests/emu/hip/hip_runtime.h (a serial fake HIP runtime)
into tests/emu/_build/libmaxsum_emu.so and driven through the same ctypes
binding.  This is test scaffolding for the GPU-less build container -- the
product never loads it; the GPU parity tests proper are in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from parity_common import check_table_updates, check_golden, compare_with_oracle, parity_cases
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params


@pytest.fixture(scope="session")
def emu_lib():
    from emu.build_emu import build
    return build()


# 0 = production (factors of a class follow their first variable: bit 7 is the default)
LAYOUTS = {"default": 0, "generic": 4, "keep_order": 8, "no_nary": 16,
           "unsorted_factors": 256, "sorted_keep_order": 128 + 8,
           "class_order_blocks": 2048, "full_width_tables": 8192, "r01_layout": 2048 + 8192}
# round-6 switches, on the cases they change only (the whole list under every layout is the GPU suite's job)
LAYOUTS_R6 = {"split_storage_types": 16777216, "no_multi_pass": 33554432}


@pytest.mark.parametrize("case", parity_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_emu_bit_exact_vs_oracle(case, dtype, emu_lib, oracle_built):
    name, make, kw = case
    compare_with_oracle(oracle_built, make(), Params(dtype=dtype, **kw), 0, lib_path=emu_lib,
                        steps=[0, 1, 1, 2, 8])


@pytest.mark.parametrize("layout", [k for k in LAYOUTS if k != "default"])
def test_emu_layout_variants(layout, emu_lib, oracle_built):
    cases = parity_cases() if layout == "unsorted_factors" else parity_cases()[:3] + parity_cases()[6:8]
    for name, make, kw in cases:
        for dtype in (("f64", "f32") if (layout == "unsorted_factors" and name in ("coloring3_soft", "mixed", "nary_meeting_d8")) else ("f64",)):
            compare_with_oracle(oracle_built, make(), Params(layout_flags=LAYOUTS[layout], dtype=dtype, **kw), 0,
                                lib_path=emu_lib, steps=[1, 6])


@pytest.mark.parametrize("layout", list(LAYOUTS_R6))
def test_emu_layout_variants_round6(layout, emu_lib, oracle_built):
    names = ("secp_small", "secp_small_m4_all", "bin2_peav_slots10", "multi_arity3_d40_max", "multi_arity4_d11", "multi_arity6_secp_m5")
    for name, make, kw in parity_cases():
        if name in names:
            compare_with_oracle(oracle_built, make(), Params(layout_flags=LAYOUTS_R6[layout], **kw), 0, lib_path=emu_lib, steps=[1, 6])


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_emu_golden(path, emu_lib):
    g, params, meta, ref_idx, ref_cost = load_golden(path)
    check_golden(g, params, meta, ref_idx, ref_cost, lib_path=emu_lib)


def test_emu_reset_and_chunked_runs(emu_lib, oracle_built):
    from pydcop_amd import generators as G
    from pydcop_amd.engine import MaxSumEngine
    g = G.random_coloring(150, seed=21)
    a = MaxSumEngine(g, Params(), lib_path=emu_lib)
    a.run(7)
    first = a.assignment()
    a.reset()
    assert a.cycle_count == 0
    for _ in range(7):
        a.run(1)
    again = a.assignment()
    np.testing.assert_array_equal(first[0], again[0])
    np.testing.assert_array_equal(first[1], again[1])


def test_emu_errors(emu_lib):
    from pydcop_amd import generators as G
    from pydcop_amd.engine import MaxSumEngine, MaxSumGpuError
    g = G.random_coloring(20, seed=1)
    bad = G.random_coloring(20, seed=1)
    bad.table_off = bad.table_off.copy()
    bad.table_off[3] += 1
    with pytest.raises(MaxSumGpuError, match="table_off"):
        MaxSumEngine(bad, Params(), lib_path=emu_lib)
    with pytest.raises(ValueError, match="damping_nodes"):
        Params(damping_nodes="sometimes").to_c()
    e = MaxSumEngine(g, Params(), lib_path=emu_lib)
    with pytest.raises(MaxSumGpuError, match="out of the domain"):
        e.eval_cost(np.full(g.n_vars, 7))
    with pytest.raises(MaxSumGpuError):
        e.run(-1)


@pytest.mark.parametrize("case", [c for c in parity_cases() if c[0] in
                                  ("coloring3_soft", "ising", "mixed_max_all", "nary_meeting_d8", "nary_mixed_dims", "bin2_coloring8_i8",
                                   "bin2_peav_slots10", "bin2_domains_to_64_int_max", "multi_arity3_d40_max", "multi_arity6_secp_m5")],
                         ids=lambda c: c[0])
def test_emu_table_updates(case, emu_lib, oracle_built):
    name, make, kw = case
    check_table_updates(oracle_built, make(), Params(**kw), lib_path=emu_lib)


def _empty_graph_checks(oracle_mod, lib_path=None):
    """Empty and degenerate inputs: no factors, no variables, invalid graphs fail loudly."""
    from pydcop_amd.engine import MaxSumGpuError
    from pydcop_amd.graph import FlatGraph
    g = FlatGraph(dom_size=[3, 2, 4], var_cost=np.arange(9.0)[::-1].copy(), factor_rowptr=[0],
                  edge_var=[], table_off=[0], tables=[], var_rowptr=[0, 0, 0, 0], var_edges=[]).validate()
    eng, ora = MaxSumEngine(g, Params(), lib_path=lib_path), oracle_mod.OracleMaxSum(g, Params())
    eng.run(5), ora.run(5)
    np.testing.assert_array_equal(eng.assignment()[0], ora.assignment()[0])
    np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1])
    assert eng.eval_cost() == ora.eval_cost() == (10.0, 0) and eng.cycle_count == 5
    g0 = FlatGraph(dom_size=[], var_cost=[], factor_rowptr=[0], edge_var=[], table_off=[0],
                   tables=[], var_rowptr=[0], var_edges=[]).validate()
    e0 = MaxSumEngine(g0, Params(), lib_path=lib_path)
    e0.run(3)
    assert e0.assignment()[0].shape == (0,) and e0.eval_cost() == (0.0, 0)
    bad = FlatGraph(dom_size=[2, 2], var_cost=np.zeros(4), factor_rowptr=[0, 2], edge_var=[0, 0],
                    table_off=[0, 4], tables=np.zeros(4), var_rowptr=[0, 2, 2], var_edges=[0, 1])
    with pytest.raises(MaxSumGpuError):  # a factor listing the same variable twice
        MaxSumEngine(bad, Params(), lib_path=lib_path)
    with pytest.raises(ValueError):
        Params(mode="sideways").to_c()


def test_emu_empty_and_invalid_graphs(emu_lib, oracle_built):
    _empty_graph_checks(oracle_built, emu_lib)


def test_emu_factor_of_arity_18(emu_lib, oracle_built):
    """The reference has no limit on a constraint's arity (maxsum.py:411-421); the engine's is what a
    2^31-entry table allows (30 binary variables).  An arity-18 factor over binary variables (262 144
    entries) among ordinary ones: generic factor kernel == oracle, bit for bit."""
    import numpy as np
    from pydcop_amd.engine import MaxSumEngine
    from pydcop_amd.generators import _finish
    rng = np.random.default_rng(18)
    n = 24
    scopes = [list(range(18))] + [[int(a), int(b)] for a, b in zip(rng.integers(0, n, 30), rng.integers(0, n, 30)) if a != b]
    tables = [rng.integers(0, 7, 2 ** len(sc)).astype(np.float64) for sc in scopes]
    rowptr = np.concatenate([[0], np.cumsum([len(sc) for sc in scopes])]).astype(np.int32)
    toff = np.concatenate([[0], np.cumsum([t.shape[0] for t in tables])]).astype(np.int64)
    g = _finish(np.full(n, 2, dtype=np.int32), rng.uniform(0, 0.01, 2 * n), rowptr,
                np.concatenate(scopes).astype(np.int32), np.concatenate(tables), toff).validate()
    p = Params(start_messages="all")
    with MaxSumEngine(g, p, lib_path=emu_lib) as eng:
        ora = oracle_built.OracleMaxSum(g, p)
        for k in (0, 1, 3):
            eng.run(k), ora.run(k)
            for x, y in zip(eng.messages(), ora.messages()):
                np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1])
        ora.close()


# ---- tiled factor order (layout.cpp; flags 131072 / 262144, default per instance) ----------

def _tiled_cases():
    from pydcop_amd import generators as G
    return [("coloring", lambda: G.random_coloring(1500, seed=3, names=False), {}),
            ("coloring_hard", lambda: G.random_coloring(900, seed=4, variant="hard", names=False), {"damping": 0.0}),
            ("mixed", lambda: [c for c in parity_cases() if c[0] == "mixed"][0][1](), {})]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_emu_tiled_factor_order_small_windows(dtype, emu_lib, oracle_built, monkeypatch):
    """Windows of 8 KB (the A/B override): a few hundred variables per bucket, so even these small
    graphs are cut into many tiles; every message is still the oracle's, bit for bit."""
    monkeypatch.setenv("MAXSUM_TILE_KB", "8")
    for name, make, kw in _tiled_cases():
        g = make()
        e = MaxSumEngine(g, Params(dtype=dtype, **kw), lib_path=emu_lib)
        if g.n_vars > 600:
            assert e.factor_order() == "tiled", name
        compare_with_oracle(oracle_built, g, Params(dtype=dtype, **kw), 0, lib_path=emu_lib, steps=[1, 6])


def test_emu_tiled_factor_order_policy(emu_lib, oracle_built):
    """Default rule: 4-byte words on a random graph larger than one window -> tiled; a grid whose
    caller order is local already -> not; flags 131072 / 262144 force it."""
    from pydcop_amd import generators as G
    g = G.random_coloring(24_000, seed=5, names=False)
    assert MaxSumEngine(g, Params(dtype="f32"), lib_path=emu_lib).factor_order() == "tiled"
    assert MaxSumEngine(g, Params(dtype="f64"), lib_path=emu_lib).factor_order() == "tiled"   # cache resident
    assert MaxSumEngine(g, Params(dtype="f32", layout_flags=262144), lib_path=emu_lib).factor_order() == "by_first_variable"
    small = G.random_coloring(2_000, seed=5, names=False)                                     # one window
    assert MaxSumEngine(small, Params(dtype="f32", layout_flags=131072), lib_path=emu_lib).factor_order() == "by_first_variable"
    grid = G.ising_grid(160, 160, seed=1, names=False)
    assert MaxSumEngine(grid, Params(dtype="f32"), lib_path=emu_lib).factor_order() == "by_first_variable"
    assert MaxSumEngine(grid, Params(dtype="f32", layout_flags=131072), lib_path=emu_lib).factor_order() == "tiled"
    compare_with_oracle(oracle_built, g, Params(dtype="f32"), 0, lib_path=emu_lib, steps=[1, 2])
    compare_with_oracle(oracle_built, grid, Params(dtype="f32", layout_flags=131072), 0, lib_path=emu_lib, steps=[2])


def test_emu_lane_grid_takes_the_binary_factors_beyond_the_register_classes(emu_lib):
    """layout.cpp: binary / unary factors with a domain of more than 4 values, or two different domain sizes, go to
    the lane-grid kernel (bin_box.h), not to the thread-per-edge generic one; flag 524288 sends them back."""
    from pydcop_amd import generators as G
    g = G.random_coloring(150, n_colors=8, seed=51)
    with MaxSumEngine(g, Params(), lib_path=emu_lib) as e:
        k = e.factor_kernels()
        assert k["lane_grid"] == g.n_factors and k["generic"] == 0, k
        assert e.table_storage()["i8"] == g.n_factors
        assert e.table_storage()["bytes_per_cycle"] == g.n_factors * 64      # 8 x 8 int8 entries, no padding
    with MaxSumEngine(g, Params(layout_flags=524288), lib_path=emu_lib) as e:
        assert e.factor_kernels()["generic"] == g.n_factors
    peav = G.peav_like(40, 25, slots=10, max_length=4, max_resources_event=4, seed=55)
    with MaxSumEngine(peav, Params(mode="max", layout_flags=16777216), lib_path=emu_lib) as e:
        k, st = e.factor_kernels(), e.table_storage()
        assert k["lane_grid"] == peav.n_factors and k["generic"] == 0, k
        assert st["full"] > 0 and st["i16"] > 0, st          # real-valued utilities; 0 / -penalty equality tables
        split_launches = e.cycle_bytes()[1]
    # default: a small group takes the wider sibling's storage type and rides in its launch (layout.cpp, BIN2_MERGE_BYTES)
    with MaxSumEngine(peav, Params(mode="max"), lib_path=emu_lib) as e:
        k, st = e.factor_kernels(), e.table_storage()
        assert k["lane_grid"] == peav.n_factors and k["generic"] == 0, k
        assert st["full"] > 0 and st["i16"] == 0 and e.cycle_bytes()[1] < split_launches, st
    three = G.random_coloring(100, seed=1)
    with MaxSumEngine(three, Params(), lib_path=emu_lib) as e:
        assert e.factor_kernels()["reg_binary"] == three.n_factors
        assert e.variable_kernels()["packed"] == three.n_vars - int((np.diff(three.var_rowptr) == 0).sum())


def test_emu_hub_variables_take_the_hub_class(emu_lib, oracle_built):
    """layout.cpp: degree above 64 on a domain of at most 8 values, above 256 or deg * D > 1024 on any, D > 256 -> K_V_HUB
    (kernels.h variable_hub: a wave per 64 outgoing edges), inside the sweep launch; flag 4194304: the round-5 classes (wide up
    to degree 256, a thread per variable beyond).  Every message the oracle's, in class order and scheduled."""
    from pydcop_amd import generators as G
    g = G.scalefree_coloring(8000, m=3, seed=5, names=False)
    deg = np.diff(g.var_rowptr)
    assert deg.max() > 256
    with MaxSumEngine(g, Params(), lib_path=emu_lib) as e:
        vk = e.variable_kernels()
        assert vk["hub"] == int((deg > 64).sum()) > 0 and vk["generic"] == 0 and vk["wide"] == 0, vk
        assert e.cycle_bytes()[1] == 1   # one launch per cycle
    with MaxSumEngine(g, Params(layout_flags=4194304), lib_path=emu_lib) as e:
        vk = e.variable_kernels()
        assert vk["hub"] == 0 and vk["generic"] == int((deg > 256).sum()) and vk["wide"] == int(((deg > 64) & (deg <= 256)).sum()), vk
    for flags in (0, 2048, 4194304, 8):
        compare_with_oracle(oracle_built, g, Params(layout_flags=flags), 0, lib_path=emu_lib, steps=[0, 1, 2, 6])
    compare_with_oracle(oracle_built, g, Params(dtype="f32", mode="max", start_messages="all"), 0, lib_path=emu_lib, steps=[0, 1, 5])
    wide_dom = G.random_mixed(6, 8, seed=23, max_arity=2, dom_choices=(300, 3))   # D = 300: beyond the wide class
    with MaxSumEngine(wide_dom, Params(), lib_path=emu_lib) as e:
        assert e.variable_kernels()["hub"] > 0 and e.variable_kernels()["generic"] == 0


def test_emu_variables_of_5_to_8_values_take_the_lane_per_edge_kernel(emu_lib, oracle_built):
    """layout.cpp: 5 <= D <= 8 and degree <= 64 -> K_V_PACK8 (k_variable_pack8), in both widths (records of 8 elements);
    flag 1048576 leaves them in the wide class; larger domains stay wide.  Every message the oracle's."""
    from pydcop_amd import generators as G
    for dtype in ("f64", "f32"):
        for colors in (5, 6, 7, 8):
            g = G.random_coloring(160, n_colors=colors, avg_degree=5, seed=70 + colors)
            connected = g.n_vars - int((np.diff(g.var_rowptr) == 0).sum())
            with MaxSumEngine(g, Params(dtype=dtype), lib_path=emu_lib) as e:
                assert e.variable_kernels()["packed8"] == connected, (dtype, colors, e.variable_kernels())
            with MaxSumEngine(g, Params(dtype=dtype, layout_flags=1048576), lib_path=emu_lib) as e:
                assert e.variable_kernels()["wide"] == connected
            compare_with_oracle(oracle_built, g, Params(dtype=dtype, start_messages="all"), 0, lib_path=emu_lib, steps=[0, 1, 2, 9])
            compare_with_oracle(oracle_built, g, Params(dtype=dtype, layout_flags=1048576), 0, lib_path=emu_lib, steps=[1, 5])
            # (default: the class's workgroups ride in the largest lane-grid factor launch; 2097152: a launch of their own)
            with MaxSumEngine(g, Params(dtype=dtype), lib_path=emu_lib) as e, \
                    MaxSumEngine(g, Params(dtype=dtype, layout_flags=2097152), lib_path=emu_lib) as e2:
                assert e.cycle_bytes()[1] + 1 == e2.cycle_bytes()[1]
            compare_with_oracle(oracle_built, g, Params(dtype=dtype, layout_flags=2097152), 0, lib_path=emu_lib, steps=[1, 5])
    mixed = G.random_mixed(60, 90, seed=77, max_arity=3, dom_choices=(2, 5, 6, 8, 9, 12))   # three variable kernels in one graph
    with MaxSumEngine(mixed, Params(), lib_path=emu_lib) as e:
        vk = e.variable_kernels()
        assert vk["packed"] > 0 and vk["packed8"] > 0 and vk["wide"] > 0, vk
    compare_with_oracle(oracle_built, mixed, Params(mode="max"), 0, lib_path=emu_lib, steps=[0, 1, 2, 9])
    nine = G.random_coloring(100, n_colors=9, seed=3)
    with MaxSumEngine(nine, Params(), lib_path=emu_lib) as e:
        assert e.variable_kernels()["packed8"] == 0 and e.variable_kernels()["wide"] > 0
