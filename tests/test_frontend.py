"""SURVEY.md section 8(f).1 -- the O(E) front-end: `factor_graph_fast` builds the SAME
computation graph as the reference's O(V*F) builder, tensorisation through the cache /
vectorised path gives the same tables as entry-by-entry evaluation, and the direct API
reproduces the reference's results (emulated engine: no GPU here).  Needs the reference
checkout (build container only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle.stage_reference import locate as _locate_reference  # noqa: E402
REF = _locate_reference() or "/root/reference"
INST = os.path.join(REF, "tests", "instances")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pydcop")),
                                reason="the pyDCOP reference checkout is not on this machine")

INSTANCES = ["graph_coloring1.yaml", "graph_coloring_tuto.yaml", "secp_simple1.yaml",
             "graph_coloring_10_4_15_0.1.yml", "graph_coloring_3agts_10vars.yaml"]


@pytest.fixture(scope="module")
def pydcop_ready():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from pydcop_amd import plugin
    plugin.install()
    return plugin


def _load(name):
    from pydcop.dcop.yamldcop import load_dcop_from_file
    return load_dcop_from_file([os.path.join(INST, name)])


@pytest.mark.parametrize("name", INSTANCES)
def test_fast_graph_equals_reference_graph(pydcop_ready, name):
    from pydcop.computations_graph import factor_graph, factor_graph_fast
    dcop = _load(name)
    ref = factor_graph.build_computation_graph(dcop)
    fast = factor_graph_fast.build_computation_graph(dcop)
    assert [n.name for n in fast.nodes] == [n.name for n in ref.nodes]
    for a, b in zip(fast.nodes, ref.nodes):
        assert type(a) is type(b) and a.type == b.type
        assert [(l.factor_node, l.variable_node) for l in a.links] == \
               [(l.factor_node, l.variable_node) for l in b.links]
        assert list(a.neighbors) == list(b.neighbors)
        assert fast.computation(a.name) is a
        assert list(fast.neighbors(a.name)) == list(ref.neighbors(a.name))
        assert list(fast.links_for_node(a.name)) == list(ref.links_for_node(a.name))
    assert fast.links == ref.links and fast.density() == ref.density()
    with pytest.raises(KeyError):
        fast.computation("no_such_node")


def test_fast_graph_argument_contract(pydcop_ready):
    from pydcop.computations_graph import factor_graph_fast
    dcop = _load("graph_coloring1.yaml")
    with pytest.raises(ValueError):
        factor_graph_fast.build_computation_graph(dcop, variables=list(dcop.variables.values()))
    with pytest.raises(ValueError):
        factor_graph_fast.build_computation_graph(None, variables=list(dcop.variables.values()))
    g = factor_graph_fast.build_computation_graph(None, variables=dcop.variables.values(),
                                                  constraints=dcop.constraints.values())
    assert len(g.nodes) == len(dcop.variables) + len(dcop.constraints)


@pytest.mark.parametrize("name", INSTANCES)
def test_tensorisation_paths_agree(pydcop_ready, name):
    """cache / vectorised evaluation == the reference's entry-by-entry evaluation."""
    import itertools
    from pydcop_amd import compile as comp
    dcop = _load(name)
    comp._TABLE_CACHE.clear()
    for _ in range(2):  # second pass is served from the cache
        for c in dcop.constraints.values():
            t = comp.tensorise_constraint(c)
            dims = list(c.dimensions)
            assert t.shape == tuple(len(v.domain) for v in dims)
            for idx in itertools.islice(itertools.product(*[range(len(v.domain)) for v in dims]), 0, 700):
                want = c(**{v.name: v.domain[i] for v, i in zip(dims, idx)})
                assert t[idx] == want


def test_vectorised_expression_and_cache_key(pydcop_ready):
    from pydcop.dcop.objects import Domain, Variable
    from pydcop.dcop.relations import constraint_from_str
    from pydcop_amd import compile as comp
    d = Domain("d", "", [0, 1, 2, 3, 4, 5])
    x, y, z, w = (Variable(n, d) for n in "xyzw")
    comp._TABLE_CACHE.clear()
    c1 = constraint_from_str("c1", "abs(x - y) * 0.5 + x", [x, y])
    t1 = comp.tensorise_constraint(c1)          # arithmetic: goes through numpy broadcasting
    pos = {v.name: i for i, v in enumerate(c1.dimensions)}
    for a in range(6):
        for b in range(6):
            idx = [0, 0]
            idx[pos["x"]], idx[pos["y"]] = a, b
            assert t1[tuple(idx)] == abs(a - b) * 0.5 + a
    n_before = len(comp._TABLE_CACHE)
    same = constraint_from_str("c1b", "abs(x - y) * 0.5 + x", [Variable("x", d), Variable("y", d)])
    assert comp.tensorise_constraint(same) is t1                     # same code + domains: cache hit
    assert len(comp._TABLE_CACHE) == n_before
    c2 = constraint_from_str("c2", "abs(z - w) * 0.5 + z", [z, w])   # other names (maybe other
    t2 = comp.tensorise_constraint(c2)                               # argument order): still right
    posz = {v.name: i for i, v in enumerate(c2.dimensions)}
    for a in range(6):
        for b in range(6):
            idx = [0, 0]
            idx[posz["z"]], idx[posz["w"]] = a, b
            assert t2[tuple(idx)] == abs(a - b) * 0.5 + a
    c3 = constraint_from_str("c3", "10 if x == y else 0", [x, y])    # does not vectorise
    t3 = comp.tensorise_constraint(c3)
    assert np.array_equal(t3, 10.0 * np.eye(6))


def test_vectorised_tables_never_trusted_where_numpy_and_python_differ(pydcop_ready):
    """int64 wrap-around and division by zero: numpy evaluates silently where Python gives the
    exact integer / raises.  The vectorised table is verified against the scalar path (every
    entry of a small table; every non-finite or > 2^53 entry of a large one)."""
    from pydcop.dcop.objects import Domain, Variable
    from pydcop.dcop.relations import constraint_from_str
    from pydcop_amd import compile as comp
    d = Domain("d", "", list(range(0, 12)))
    x, y = Variable("x", d), Variable("y", d)
    big = constraint_from_str("big", "x ** 30 + y", [x, y])        # 11 ** 30 wraps in int64
    t = comp.tensorise_constraint(big)
    posx = [v.name for v in big.dimensions].index("x")
    idx = [3, 3]
    idx[posx] = 11
    assert t[tuple(idx)] == float(11 ** 30 + 3)
    div = constraint_from_str("div", "x / y", [x, y])              # y = 0 is in the domain
    with pytest.raises(ZeroDivisionError):
        comp.tensorise_constraint(div)
    # large table (> 4096 entries): the suspect entries are checked, not just 16 samples
    d3 = Domain("d3", "", list(range(0, 20)))
    a, b, c = Variable("a", d3), Variable("b", d3), Variable("c", d3)
    div3 = constraint_from_str("div3", "a / (b - 7) + c", [a, b, c])
    with pytest.raises(ZeroDivisionError):
        comp.tensorise_constraint(div3)
    # unhashable domain values: no cache key, still tensorised
    dl = Domain("dl", "", [[0], [1]])
    u = Variable("u", dl)
    cu = constraint_from_str("cu", "len(u) + u[0]", [u])
    assert np.array_equal(comp.tensorise_constraint(cu), np.array([1.0, 2.0]))


@pytest.mark.parametrize("name,expected,cost", [
    ("graph_coloring1.yaml", {"v1": "R", "v2": "G", "v3": "R"}, -0.1),
    ("secp_simple1.yaml", {"l1": 0, "l2": 3, "l3": 4, "m1": 3}, None),
    ("graph_coloring_tuto.yaml", None, 12),
])
def test_direct_api_reproduces_reference_results(pydcop_ready, name, expected, cost):
    """SURVEY.md section 8c golden results through pydcop_amd.api (emulated engine)."""
    from emu.build_emu import build
    from pydcop_amd.api import solve_yaml
    res = solve_yaml(os.path.join(INST, name), cycles=20, noise=0, lib_path=build(),
                     cost_every=5, infinity=float("inf"))
    if expected is not None:
        assert res["assignment"] == expected
    if cost is not None:
        assert abs(res["cost"] - cost) < 1e-9
    assert res["violation"] == 0 and res["cycle"] == 20
    assert [c[0] for c in res["cost_curve"]] == [5, 10, 15, 20]
    assert abs(res["cost_curve"][-1][1] - res["cost"]) < 1e-9   # device cost == DCOP.solution_cost


@pytest.mark.parametrize("name", ["graph_coloring1.yaml", "secp_simple1.yaml"])
def test_yaml_export_to_instance_file(pydcop_ready, name, tmp_path, capsys):
    """`python -m pydcop_amd.api --export`: YAML -> .npz instance, solved without pyDCOP
    objects, equals the solve from the YAML."""
    import json
    from emu.build_emu import build
    from pydcop_amd import api
    from pydcop_amd.graph import FlatGraph
    out = str(tmp_path / "inst.npz")
    api.main(["--export", out, "-p", "noise:0", os.path.join(INST, name)])
    info = json.loads(capsys.readouterr().out)
    g, header = FlatGraph.load(out)
    assert info["n_vars"] == g.n_vars and header["objective"] == info["objective"]
    assert header["meta"]["source"] == [name]
    a = api.solve_yaml(os.path.join(INST, name), cycles=15, noise=0, lib_path=build(), infinity=float("inf"))
    b = api.solve_flat(g, header["objective"], 15, lib_path=build(), infinity=float("inf"))
    assert a["assignment"] == b["assignment"] and a["violation"] == b["violation"]
    assert abs(a["cost"] - b["cost"]) < 1e-9


@pytest.mark.parametrize("name", ["graph_coloring1.yaml", "secp_simple1.yaml"])
def test_noise_never_reaches_the_reported_cost(pydcop_ready, name, tmp_path, capsys):
    """The Max-Sum noise (maxsum.py:476-487) breaks ties inside the computations only: the
    cost of an instance exported WITH noise (the default, 0.01) is DCOP.solution_cost of
    the selected values (dcop.py:308-367), on the device as on the host, curve included."""
    from emu.build_emu import build
    from pydcop_amd import api
    from pydcop_amd.graph import FlatGraph
    from pydcop.dcop.yamldcop import load_dcop_from_file
    out = str(tmp_path / "noisy.npz")
    api.main(["--export", out, "-p", "noise:0.5", "-p", "seed:3", os.path.join(INST, name)])
    capsys.readouterr()
    g, header = FlatGraph.load(out)
    assert g.eval_var_cost is not None and (g.var_cost != g.eval_var_cost).any()
    res = api.solve_flat(g, header["objective"], 12, lib_path=build(), infinity=float("inf"), cost_every=4)
    dcop = load_dcop_from_file([os.path.join(INST, name)])
    violation, cost = dcop.solution_cost(res["assignment"], float("inf"))
    assert res["violation"] == violation and abs(res["cost"] - cost) < 1e-9
    assert abs(res["cost_curve"][-1][1] - cost) < 1e-9
    # same through the YAML path
    res2 = api.solve_yaml(os.path.join(INST, name), cycles=12, noise=0.5, seed=3, lib_path=build(),
                          infinity=float("inf"), cost_every=4)
    assert res2["assignment"] == res["assignment"] and abs(res2["cost_curve"][-1][1] - res2["cost"]) < 1e-9


def test_plugin_uses_fast_graph_when_asked(pydcop_ready):
    from pydcop.algorithms import load_algorithm_module
    mod = load_algorithm_module("maxsum_gpu")
    old = mod.GRAPH_TYPE
    try:
        pydcop_ready.install(fast_graph=True)
        assert mod.GRAPH_TYPE == "factor_graph_fast"
        from importlib import import_module
        gm = import_module("pydcop.computations_graph." + mod.GRAPH_TYPE)
        assert hasattr(gm, "build_computation_graph")
    finally:
        mod.GRAPH_TYPE = old


@pytest.mark.parametrize("name", ["graph_coloring1.yaml", "graph_coloring_3agts_10vars.yaml", "secp_simple1.yaml"])
def test_direct_api_local_search_equals_the_reference(pydcop_ready, name):
    """`solve_yaml(algo="mgm" / "dsa")` = the reference's own MgmComputation / DsaComputation objects
    after the same number of rounds (DSA: both sides on the keyed generator, indexed by the compiled
    graph's variable order)."""
    from emu.build_emu import build
    from oracle import ref_harness
    from pydcop_amd import api
    from pydcop.dcop.yamldcop import load_dcop_from_file
    path = os.path.join(INST, name)
    dcop = load_dcop_from_file([path])
    values, _, _ = ref_harness.run_reference_mgm(dcop, 6)
    res = api.solve_yaml(path, cycles=6, algo="mgm", lib_path=build(), infinity=float("inf"))
    assert res["assignment"] == values
    graph = api.compile_dcop(dcop, noise=0.0)
    index = {n: i for i, n in enumerate(graph.var_names)}
    for variant in ("A", "C"):
        dcop = load_dcop_from_file([path])
        try:
            values, _, _ = ref_harness.run_reference_dsa(dcop, 7, variant=variant, probability=0.6, seed=5, var_index=index)
        except ZeroDivisionError:   # p_mode / arity corner of the reference on variables without binary constraint
            continue
        res = api.solve_yaml(path, cycles=7, algo="dsa", variant=variant, probability=0.6, seed=5,
                             lib_path=build(), infinity=float("inf"))
        assert res["assignment"] == values, variant


@pytest.mark.parametrize("name", ["graph_coloring1.yaml", "graph_coloring_tuto.yaml", "secp_simple1.yaml"])
def test_direct_api_amaxsum_equals_the_reference(pydcop_ready, name):
    """`solve_yaml(algo="amaxsum")` = the reference's amaxsum computations under FIFO delivery after
    the same number of generations."""
    from emu.build_emu import build
    from oracle import ref_harness
    from pydcop_amd import api
    from pydcop.dcop.yamldcop import load_dcop_from_file
    path = os.path.join(INST, name)
    values, _, info = ref_harness.run_reference_amaxsum(load_dcop_from_file([path]), 8)
    res = api.solve_yaml(path, cycles=8, algo="amaxsum", noise=0, lib_path=build(), infinity=float("inf"))
    assert res["assignment"] == values
