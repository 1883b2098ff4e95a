"""Cases of the plug-in driven through the pyDCOP STAND-IN (tests/standin: a mock of what an agent
does to a computation): `pydcop_amd/algorithms/maxsum_gpu.py` -- build_computation, the proxies,
the session that compiles the registered ComputationDefs and owns the engine.  Not collected by
pytest: the stand-in must never share a process with the real pyDCOP (which the other tests import,
on the GPU box too since round 3), so tests/test_gpu_plugin.py (real libmaxsum_hip.so, MI355X) and
tests/test_plugin_standin.py (emulated engine) run these functions in a child process.  The real
orchestrator + agents + HIP library meet in tests/test_gpu_reference_e2e.py.

Expected results are the reference's own (tests/dcop_cli/test_solve.py:100-130: v1=R, v2=G, v3=R)
and, for the seeded instances, the oracle on the arrays the session compiled."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def load_plugin():
    sys.path.insert(0, os.path.join(HERE, "standin"))
    import pydcop
    assert pydcop.STANDIN
    from pydcop_amd import plugin
    plugin.install()
    from pydcop.algorithms import load_algorithm_module
    return load_algorithm_module("maxsum_gpu")


class Var:
    def __init__(self, name, domain, cost=None, initial_value=None):
        self.name, self.domain, self._cost, self.initial_value = name, list(domain), cost, initial_value

    def cost_for_val(self, val):
        return self._cost(val) if self._cost else 0.0


class Table:
    """Extensional constraint: `_m` like NAryMatrixRelation (relations.py:716-733)."""
    def __init__(self, name, dimensions, m):
        self.name, self.dimensions, self._m = name, list(dimensions), np.asarray(m, dtype=np.float64)

    def __call__(self, **kw):
        return float(self._m[tuple(v.domain.index(kw[v.name]) for v in self.dimensions)])


class Intention:
    def __init__(self, name, dimensions, fn):
        self.name, self.dimensions, self._fn = name, list(dimensions), fn

    def __call__(self, **kw):
        return self._fn(**kw)


def _nodes(variables, constraints):
    from pydcop.computations_graph.factor_graph import FactorComputationNode, VariableComputationNode
    links = {v.name: [] for v in variables}
    for c in constraints:
        for v in c.dimensions:
            links[v.name].append(c.name)
    return ([VariableComputationNode(v, links[v.name]) for v in variables],
            [FactorComputationNode(c) for c in constraints])


def _solve_through_plugin(plugin_mod, variables, constraints, mode="min", algo_name="maxsum_gpu", **params):
    """What run_local_thread_dcop + one agent do, on this thread: one computation per node
    through build_computation, start them all, serve their periodic actions until every one
    has called finished()."""
    from pydcop.algorithms import AlgorithmDef, ComputationDef
    from pydcop.infrastructure.computations import MiniAgent
    algo = AlgorithmDef.build_with_default_param(algo_name, params, mode=mode,
                                                 parameters_definitions=plugin_mod.algo_params)
    vnodes, fnodes = _nodes(variables, constraints)
    comps = [plugin_mod.build_computation(ComputationDef(n, algo)) for n in vnodes + fnodes]
    session = comps[0]._session
    agent = MiniAgent(comps)
    agent.start_all()
    agent.pump(lambda: all(c.is_finished for c in comps))
    values = {c.name: (c.current_value, c.current_cost) for c in comps if hasattr(c, "current_value")}
    cycles = {c.cycle_count for c in comps}
    graph = session.graph
    agent.stop_all()
    assert session.engine is None and session.stopped  # closed by the LAST proxy's stop
    return values, cycles, graph, comps


def _standin_in_use():
    import pydcop
    return getattr(pydcop, "STANDIN", False)


def test_native_library_is_the_hip_build(plugin_mod):
    from pydcop_amd.engine import DEFAULT_LIB, load_library
    lib = load_library()
    assert lib.mxs_build_kind() == 1 and os.path.basename(DEFAULT_LIB).startswith("libmaxsum_hip")
    with open("/proc/self/maps") as f:
        assert any("libmaxsum_hip" in line for line in f)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_graph_coloring1_through_the_proxies(plugin_mod, precision):
    v1 = Var("v1", "RG", lambda x: -0.1 if x == "R" else 0.1)
    v2 = Var("v2", "RG", lambda x: -0.1 if x == "G" else 0.1)
    v3 = Var("v3", "RG", lambda x: -0.1 if x == "G" else 0.1)
    d12 = Intention("diff_1_2", [v1, v2], lambda v1, v2: 1 if v1 == v2 else 0)
    d23 = Intention("diff_2_3", [v3, v2], lambda v3, v2: 1 if v3 == v2 else 0)
    values, cycles, graph, comps = _solve_through_plugin(
        plugin_mod, [v1, v2, v3], [d12, d23], stop_cycle=20, noise=0, precision=precision)
    assert {k: v[0] for k, v in values.items()} == {"v1": "R", "v2": "G", "v3": "R"}
    assert cycles == {20}
    if _standin_in_use():
        assert all(c.cycle_events and c.cycle_events[-1] == 20 for c in comps)  # cycle events fired


def test_random_coloring_through_the_proxies_equals_oracle(plugin_mod, oracle_built):
    """300 variables / 600 extensional constraints registered node by node (as the reference
    hands them over), noise seeded: the values the proxies publish are the oracle's on the
    arrays the session compiled."""
    from pydcop_amd.graph import Params
    rng = np.random.default_rng(5)
    n, m = 300, 600
    vs = [Var(f"x{i:03d}", [0, 1, 2], (lambda i: (lambda d: 0.001 * ((d + i) % 3)))(i)) for i in range(n)]
    cons, seen = [], set()
    while len(cons) < m:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a == b or (min(a, b), max(a, b)) in seen:
            continue
        seen.add((min(a, b), max(a, b)))
        cons.append(Table(f"c{len(cons):03d}", [vs[a], vs[b]], rng.integers(0, 10, (3, 3))))
    values, cycles, graph, _ = _solve_through_plugin(plugin_mod, vs, cons, stop_cycle=25, noise=0.01, seed=4)
    assert cycles == {25} and graph.eval_var_cost is not None
    ora = oracle_built.OracleMaxSum(graph, Params())
    ora.run(25)
    idx, belief = ora.assignment()
    for i, name in enumerate(graph.var_names):
        assert values[name][0] == graph.domains[i][int(idx[i])]
        assert values[name][1] == belief[i]


def test_change_factor_function_and_stop_of_one_proxy(plugin_mod):
    """maxsum_dynamic's change_factor_function through the factor proxy while the engine keeps
    sweeping (stop_cycle 0), and: stopping ONE computation leaves the others their engine."""
    from pydcop.algorithms import AlgorithmDef, ComputationDef
    from pydcop.infrastructure.computations import MiniAgent
    a, b = Var("a", [0, 1]), Var("b", [0, 1])
    eq = Table("eq", [a, b], [[0, 5], [5, 0]])           # prefers a == b
    pa = Table("pa", [a], [0.0, 1.0])                    # a = 0
    algo = AlgorithmDef.build_with_default_param("maxsum_gpu", {"noise": 0, "chunk": 5}, mode="min",
                                                 parameters_definitions=plugin_mod.algo_params)
    vnodes, fnodes = _nodes([a, b], [eq, pa])
    comps = {n.name: plugin_mod.build_computation(ComputationDef(n, algo)) for n in vnodes + fnodes}
    agent = MiniAgent(comps.values())
    agent.start_all()
    session = comps["a"]._session
    agent.pump(lambda: session.cycles >= 20)
    agent.pump(lambda: comps["b"].cycle_count >= 20)
    assert (comps["a"].current_value, comps["b"].current_value) == (0, 0)
    comps["eq"].change_factor_function(Table("eq", [a, b], [[5, 0], [0, 5]]))  # now prefers a != b
    at = session.cycles
    agent.pump(lambda: session.cycles >= at + 30 and comps["b"].cycle_count >= at + 30)
    assert (comps["a"].current_value, comps["b"].current_value) == (0, 1)
    # a new function over OTHER variables (maxsum_dynamic.py:234-271): `eq` lets go of a
    comps["eq"].change_factor_function(Table("eq", [b], [0.0, 3.0]))            # now: b = 0
    at = session.cycles
    agent.pump(lambda: session.cycles >= at + 30 and comps["b"].cycle_count >= at + 30)
    assert (comps["a"].current_value, comps["b"].current_value) == (0, 0)
    assert session.engine.relayouts == 1 and session.graph.n_edges == 2
    comps["pa"].stop()                                   # e.g. an agent removal
    assert session.engine is not None and not session.stopped
    at = session.cycles
    agent.pump(lambda: session.cycles >= at + 10)
    agent.stop_all()
    assert session.engine is None and session.stopped


def test_amaxsum_gpu_through_the_proxies_equals_oracle(plugin_mod, oracle_built):
    """`--algo amaxsum_gpu`: the asynchronous schedule (FIFO generations) through the same proxies;
    runs until no message is left (stop_cycle 0) and publishes what the oracle holds at quiescence."""
    from pydcop.algorithms import load_algorithm_module
    from oracle.amaxsum_oracle import OracleAMaxSum
    from pydcop_amd.graph import Params
    amod = load_algorithm_module("amaxsum_gpu")
    rng = np.random.default_rng(9)
    n, m = 40, 70
    vs = [Var(f"y{i:02d}", [0, 1, 2], (lambda i: (lambda d: 0.001 * ((d + 2 * i) % 3)))(i)) for i in range(n)]
    cons, seen = [], set()
    while len(cons) < m:
        a, b = (int(x) for x in rng.integers(0, n, 2))
        if a == b or (min(a, b), max(a, b)) in seen:
            continue
        seen.add((min(a, b), max(a, b)))
        cons.append(Table(f"k{len(cons):02d}", [vs[a], vs[b]], rng.integers(0, 10, (3, 3))))
    values, cycles, graph, _ = _solve_through_plugin(amod, vs, cons, algo_name="amaxsum_gpu", noise=0,
                                                     start_messages="leafs_vars", chunk=8)
    ora = OracleAMaxSum(graph, Params(start_messages="leafs_vars"))
    ora.run(-1)
    assert ora.pending == 0 and cycles == {ora.generation + 1}
    idx, belief = ora.assignment()
    for i, name in enumerate(graph.var_names):
        assert values[name] == (graph.domains[i][int(idx[i])], belief[i])
