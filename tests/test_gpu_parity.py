"""GPU parity tests proper: the HIP engine (pydcop_amd/csrc/libmaxsum_hip.so,
through the C-ABI) against the CPU oracle on the same seeded inputs, against the
golden vectors generated from the reference, and -- at BASELINE.json's full sizes
-- through size-independent properties.

Bar: op-for-op identical arithmetic, so f64 and f32 are compared BIT-EXACT with
the oracle's f64/f32 builds; against the reference's golden vectors the
north-star tolerance (1e-5) applies.
"""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from parity_common import check_table_updates, check_golden, compare_with_oracle, parity_cases
from pydcop_amd import generators as G
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", parity_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_bit_exact_vs_oracle(case, dtype, oracle_built):
    name, make, kw = case
    compare_with_oracle(oracle_built, make(), Params(dtype=dtype, **kw), 0,
                        steps=[0, 1, 1, 2, 8, 30])


@pytest.mark.parametrize("flags", [4, 8, 16, 4 + 8 + 16, 128, 256, 128 + 8, 2048, 2048 + 256, 4096 + 8, 8192, 8192 + 2048,
                                   16777216, 33554432])  # (round 6: storage types of a lane-grid shape kept apart; no multi-pass workgroup kernel)
def test_layout_variants(flags, oracle_built):
    for name, make, kw in parity_cases():
        if (flags & 4) and name in ("hub_deg1100_max_all", "hub_deg3400"):
            continue   # (generic kernels only: ONE thread walks such a hub's O(deg^2 * D) chain -- minutes; degree 513 stays in)
        compare_with_oracle(oracle_built, make(), Params(layout_flags=flags, **kw), 0, steps=[1, 9])


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_reference_vectors(path):
    g, params, meta, ref_idx, ref_cost = load_golden(path)
    check_golden(g, params, meta, ref_idx, ref_cost)
    check_golden(g, params, meta, ref_idx, ref_cost, layout_flags=4)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_tiled_factor_order(dtype, oracle_built, monkeypatch):
    """The binary factors in tiled order (layout.cpp): default rule on the bench instance, forced on
    an Ising grid, forced off, and with 8 KB windows on small graphs -- always the oracle's messages."""
    from pydcop_amd.engine import MaxSumEngine
    g = G.random_coloring(100_000, seed=0, names=False)
    assert MaxSumEngine(g, Params(dtype=dtype)).factor_order() == "tiled"
    compare_with_oracle(oracle_built, g, Params(dtype=dtype), 0, steps=[1, 7])
    assert MaxSumEngine(g, Params(dtype=dtype, layout_flags=262144)).factor_order() == "by_first_variable"
    compare_with_oracle(oracle_built, g, Params(dtype=dtype, layout_flags=262144), 0, steps=[3])
    grid = G.ising_grid(200, 200, seed=2, names=False)
    assert MaxSumEngine(grid, Params(dtype=dtype)).factor_order() == "by_first_variable"
    compare_with_oracle(oracle_built, grid, Params(dtype=dtype, layout_flags=131072), 0, steps=[1, 5])
    monkeypatch.setenv("MAXSUM_TILE_KB", "8")
    for name, make, kw in parity_cases():
        compare_with_oracle(oracle_built, make(), Params(dtype=dtype, **kw), 0, steps=[1, 9])


def test_config2_10k_coloring(oracle_built):
    """BASELINE.json configs[1]: random 3-colouring, 10k vars, degree 4."""
    for variant in ("soft", "hard"):
        g = G.random_coloring(10_000, seed=0, variant=variant, names=False)
        compare_with_oracle(oracle_built, g, Params(), 0, steps=[1, 49])


def test_north_star_100k_coloring(oracle_built):
    """The bench workload: 100k vars / 200k factors / 400k edges, 40 cycles."""
    g = G.random_coloring(100_000, seed=0, names=False)
    compare_with_oracle(oracle_built, g, Params(), 0, steps=[40])
    compare_with_oracle(oracle_built, g, Params(dtype="f32"), 0, steps=[10])


def test_ising_grid_256(oracle_built):
    g = G.ising_grid(256, 256, seed=0, names=False)
    compare_with_oracle(oracle_built, g, Params(), 0, steps=[20])


# BASELINE.json configs[2..4] at their stated sizes, bit-exact against the oracle (OpenMP over
# factors / variables; the order inside a message is the reference's either way): 32-bit
# offsets, block_base[] class tables, multi-generation grids and the 24^3 launch grouping of
# k_factor_nary only show at full size.
# Long enough that the state machine of the send rule is compared at full size too: edges that were
# "sent again" three times go silent (counters frozen at SAME_COUNT, the receiver keeps the old
# message -- maxsum.py:371-377); `compare_with_oracle(expect_silent=True)` asserts that the run got there.
_FULL = {
    "ising_1024": (lambda: G.ising_grid(1024, 1024, seed=0, names=False), "min", [1, 5, 34]),
    "coloring_1m_deg6": (lambda: G.random_coloring(1_000_000, avg_degree=6, n_colors=3, seed=0, names=False),
                         "min", [1, 5, 34]),
    "meeting_50k": (lambda: G.meeting_like(50_000, dom=24, arity=3, seed=0, names=False), "max", [1, 3, 26]),
    # round 5: what the reference's own generators emit between "tiny" and "huge" -- the lane-grid kernel (bin_box.h)
    "peav_50k": (lambda: G.peav_like(seed=0, names=False), "max", [1, 3, 26]),
    "coloring_100k_d8": (lambda: G.random_coloring(100_000, avg_degree=4, n_colors=8, seed=0, names=False), "min", [1, 5, 34]),
    # configs[4] with real-valued utilities: the workgroup-per-factor kernel on FULL-WIDTH 24^3 tables
    "meeting_50k_float": (lambda: G.meeting_like(50_000, dom=24, arity=3, seed=0, names=False, float_tables=True), "max", [1, 3, 22]),
    # configs[4] over 18..24 slots per variable: every table on a lane grid that overhangs it (box records, round 5)
    "meeting_50k_hetero": (lambda: G.meeting_hetero(50_000, doms=(24, 23, 22, 21, 20, 19, 18), arity=3, seed=0, names=False), "max", [1, 3, 26]),
    # round 6: what `--graph scalefree` emits (graphcoloring.py:322-340) -- hub variables of degree up to ~700 (100k) / ~2 200 (1M):
    # the wave-per-64-edges class (kernels.h variable_hub) riding in the sweep launch, one tile and several per value of d
    "coloring_100k_scalefree": (lambda: G.scalefree_coloring(100_000, m=2, n_colors=3, seed=0, names=False), "min", [1, 5, 34]),
    "coloring_1m_scalefree": (lambda: G.scalefree_coloring(1_000_000, m=2, n_colors=3, seed=0, names=False), "min", [1, 5, 34]),
    # round 6: the reference's `generate secp` at 100k variables (generators.secp_like: secp.py's expressions table for table);
    # _m4 = --max_model_size 4: model constraints of arity 5 on the workgroup-per-factor kernels
    "secp_100k": (lambda: G.secp_like(60_000, 40_000, 50_000, max_model_size=3, seed=0, names=False), "min", [1, 5, 34]),
    "secp_100k_m4": (lambda: G.secp_like(60_000, 40_000, 50_000, max_model_size=4, seed=0, names=False), "min", [1, 5, 20]),
    # round 6: the workgroup-per-factor kernel in passes (k_factor_nary<.., MULTI>): SECP with --max_model_size 5 (arity 6, 15 625
    # entries) and configs[4]'s model over 40 slots (64 000-entry tables: 1 600 entries per value of the first variable)
    "secp_30k_m5": (lambda: G.secp_like(18_000, 12_000, 15_000, max_model_size=5, seed=0, names=False), "min", [1, 5, 20]),
    "meeting_5k_d40": (lambda: G.meeting_like(5_000, dom=40, arity=3, seed=0, names=False), "max", [1, 3, 22]),
}
_full_cache = {}


def _full_graph(name):
    if name not in _full_cache:
        _full_cache.clear()  # one full-size instance in memory at a time (meeting_50k: 5.5 GB of tables)
        _full_cache[name] = _FULL[name][0]()
    return _full_cache[name]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", list(_FULL))
def test_full_size_bit_exact_vs_oracle(name, dtype, oracle_built):
    _, mode, steps = _FULL[name]
    threads = min(64, len(os.sched_getaffinity(0)))
    # (f32 -- narrower than the reference's arithmetic, an extra -- keeps the short run)
    compare_with_oracle(oracle_built, _full_graph(name), Params(mode=mode, dtype=dtype), 0,
                        steps=steps if dtype == "f64" else steps[:2], threads=threads,
                        expect_silent=dtype == "f64")
    if dtype == "f32":
        _full_cache.clear()


def test_scalefree_hubs_take_the_hub_class():
    """No variable of a scale-free colouring is left to the thread-per-variable kernel (one thread walked a hub's
    O(deg^2 * D) chain); flag 4194304 restores that for A/B runs."""
    g = _full_graph("coloring_100k_scalefree")
    deg = np.diff(g.var_rowptr)
    with MaxSumEngine(g, Params()) as e:
        vk = e.variable_kernels()
        assert vk["generic"] == 0 and vk["wide"] == 0 and vk["hub"] == int((deg > 64).sum()) and e.cycle_bytes()[1] == 1, vk
    with MaxSumEngine(g, Params(layout_flags=4194304)) as e:
        vk = e.variable_kernels()
        assert vk["hub"] == 0 and vk["generic"] == int((deg > 256).sum()), vk


def test_graph_replay_equals_eager(oracle_built):
    """hipGraph replay of the cycle loop gives the same state as eager launches."""
    g = G.random_coloring(5000, seed=3, names=False)
    a = MaxSumEngine(g, Params(graph_chunk=8))
    b = MaxSumEngine(g, Params(graph_chunk=0))
    a.run(1), b.run(1)          # odd parity first
    a.run(100), b.run(100)
    a.run(37), b.run(37)
    for x, y in zip(a.messages(), b.messages()):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a.assignment()[0], b.assignment()[0])
    assert a.cycle_count == b.cycle_count == 138


def test_full_size_properties():
    """Size-independent properties on the full north-star instance (no oracle):
    determinism, reset idempotence, relabelling invariance, cost bookkeeping."""
    g = G.random_coloring(100_000, seed=7, names=False)
    e1, e2 = MaxSumEngine(g, Params()), MaxSumEngine(g, Params())
    e1.run(60), e2.run(25), e2.run(35)
    i1, b1 = e1.assignment()
    i2, b2 = e2.assignment()
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(b1, b2)
    e2.reset(), e2.run(60)
    np.testing.assert_array_equal(e2.assignment()[0], i1)
    # solution cost on the device == numpy evaluation of the same assignment
    t = g.tables.reshape(-1, 3, 3)
    a0, a1 = i1[g.edge_var[0::2]], i1[g.edge_var[1::2]]
    want = t[np.arange(t.shape[0]), a0, a1].sum() + g.var_cost.reshape(-1, 3)[np.arange(g.n_vars), i1].sum()
    got, viol = e1.eval_cost()
    assert viol == 0 and abs(got - want) <= 1e-9 * abs(want)
    # Max-Sum must beat the trivial start assignment by a wide margin
    e3 = MaxSumEngine(g, Params())
    start_cost, _ = e3.eval_cost()
    assert got < 0.75 * start_cost
    # relabelling the factors (reversed order) only changes floating-point
    # summation order on the variable side: same assignment up to near-ties
    nf = g.n_factors
    perm = np.arange(nf)[::-1]
    ev = g.edge_var.reshape(nf, 2)[perm].reshape(-1)
    vr, ve = g.var_side_from_edges(ev, g.n_vars)
    from pydcop_amd.graph import FlatGraph
    g2 = FlatGraph(dom_size=g.dom_size, var_cost=g.var_cost, factor_rowptr=g.factor_rowptr,
                   edge_var=ev, table_off=g.table_off, tables=t[perm].reshape(-1),
                   var_rowptr=vr, var_edges=ve)
    e4 = MaxSumEngine(g2, Params())
    e4.run(60)
    assert (e4.assignment()[0] != i1).mean() < 1e-3


def test_max_mode_is_negated_min_mode():
    g = G.random_coloring(3000, seed=5, names=False)
    from pydcop_amd.graph import FlatGraph
    neg = FlatGraph(dom_size=g.dom_size, var_cost=-g.var_cost, factor_rowptr=g.factor_rowptr,
                    edge_var=g.edge_var, table_off=g.table_off, tables=-g.tables,
                    var_rowptr=g.var_rowptr, var_edges=g.var_edges)
    a, b = MaxSumEngine(g, Params(mode="min")), MaxSumEngine(neg, Params(mode="max"))
    a.run(30), b.run(30)
    np.testing.assert_array_equal(a.assignment()[0], b.assignment()[0])
    np.testing.assert_array_equal(a.assignment()[1], -b.assignment()[1])


@pytest.mark.parametrize("case", [c for c in parity_cases() if c[0] in
                                  ("coloring3_soft", "ising", "mixed_max_all", "nary_meeting_d8", "nary_mixed_dims", "bin2_coloring8_i8",
                                   "bin2_peav_slots10", "bin2_domains_to_64_int_max", "multi_arity3_d40_max", "multi_arity6_secp_m5")],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_table_updates(case, dtype, oracle_built):
    """change_factor_function (maxsum_dynamic.py:80-104) through mxs_update_factor_table."""
    name, make, kw = case
    check_table_updates(oracle_built, make(), Params(dtype=dtype, **kw))


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


def test_empty_and_invalid_graphs(oracle_built):
    _empty_graph_checks(oracle_built)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_dynamic_maxsum(dtype, oracle_built):
    """maxsum_dynamic on the device: table swaps, scope changes (re-layout + mxs_set_state),
    external values sliced by mxs_slice_factor, checkpoint / resume -- bit-exact vs the oracle."""
    from dynamic_common import check_dynamic_run
    for seed in (1, 2):
        check_dynamic_run(oracle_built, dtype=dtype, seed=seed)
    check_dynamic_run(oracle_built, dtype=dtype, seed=3, float_tables=False)


def test_factor_of_arity_18(oracle_built):
    """The reference has no arity limit (maxsum.py:411-421); the engine's is 30 (a 2^31-entry table).  An
    arity-18 factor over binary variables among ordinary ones: generic factor kernel == oracle, bit for bit."""
    import test_emu_engine as E
    E.test_emu_factor_of_arity_18(None, oracle_built)
