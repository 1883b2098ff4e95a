"""The plugin's data path on a real MI355X, without pyDCOP (which is not on the GPU box):
duck-typed variable / constraint / computation-node objects with the attributes
`pydcop_amd.compile.compile_nodes` reads from the reference's classes
(pydcop/dcop/objects.py:175, pydcop/dcop/relations.py:456,672,
pydcop/computations_graph/factor_graph.py:45,104) -> flat arrays -> HIP engine -> values.
The instance is the reference's tests/instances/graph_coloring1.yaml restated inline; the
expected result is the reference's own (tests/dcop_cli/test_solve.py:100-130: v1=R, v2=G,
v3=R, cost -0.1)."""
import numpy as np
import pytest

from pydcop_amd.compile import assignment_to_values, compile_nodes
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


class Var:
    def __init__(self, name, domain, cost, initial_value=None):
        self.name, self.domain, self._cost, self.initial_value = name, list(domain), cost, initial_value

    def cost_for_val(self, val):
        return self._cost(val)


class Intention:
    """An intentional constraint: called with keyword arguments like NAryFunctionRelation."""
    def __init__(self, name, dimensions, fn):
        self.name, self.dimensions, self._fn = name, list(dimensions), fn

    def __call__(self, **kw):
        return self._fn(**kw)


class Link:
    def __init__(self, factor_node, variable_node):
        self.factor_node, self.variable_node = factor_node, variable_node


class VarNode:
    type = "VariableComputation"

    def __init__(self, variable, factor_names):
        self.name, self.variable = variable.name, variable
        self.links = [Link(f, variable.name) for f in factor_names]


class FactorNode:
    type = "FactorComputation"

    def __init__(self, factor):
        self.name, self.factor = factor.name, factor


def graph_coloring1():
    v1 = Var("v1", "RG", lambda x: -0.1 if x == "R" else 0.1)
    v2 = Var("v2", "RG", lambda x: -0.1 if x == "G" else 0.1)
    v3 = Var("v3", "RG", lambda x: -0.1 if x == "G" else 0.1)
    d12 = Intention("diff_1_2", [v1, v2], lambda v1, v2: 1 if v1 == v2 else 0)
    d23 = Intention("diff_2_3", [v3, v2], lambda v3, v2: 1 if v3 == v2 else 0)
    vnodes = [VarNode(v1, ["diff_1_2"]), VarNode(v2, ["diff_1_2", "diff_2_3"]), VarNode(v3, ["diff_2_3"])]
    return vnodes, [FactorNode(d12), FactorNode(d23)]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("cycles", [5, 20, 50])
def test_graph_coloring1_through_compile_nodes(dtype, cycles):
    vnodes, fnodes = graph_coloring1()
    g = compile_nodes(vnodes, fnodes, noise=0.0)
    assert g.n_vars == 3 and g.n_factors == 2 and g.n_edges == 4
    with MaxSumEngine(g, Params(mode="min", dtype=dtype)) as eng:
        eng.run(cycles)
        idx, _ = eng.assignment()
        cost, viol = eng.eval_cost()
    assert assignment_to_values(g, idx) == {"v1": "R", "v2": "G", "v3": "R"}
    assert viol == 0 and abs(cost - (-0.1)) < 1e-6


def test_intentional_arity3_and_initial_values(oracle_built):
    """A D=5 arity-3 intentional factor (generic / wide classes), initial values and seeded
    noise through the same path, against the oracle on the compiled arrays."""
    from parity_common import compare_with_oracle
    dom = [0, 1, 2, 3, 4]
    vs = [Var(f"x{i}", dom, (lambda i: (lambda x: 0.01 * ((x + i) % 5)))(i), initial_value=(i % 5 if i % 3 == 0 else None))
          for i in range(12)]
    facs, links = [], {v.name: [] for v in vs}
    rng = np.random.default_rng(3)
    for k in range(10):
        a, b, c = rng.choice(12, size=3, replace=False)
        f = Intention(f"c{k:02d}", [vs[a], vs[b], vs[c]],
                      (lambda n: (lambda **kw: float(abs(sum(kw.values()) - n))))(k))
        facs.append(f)
        for v in f.dimensions:
            links[v.name].append(f.name)
    g = compile_nodes([VarNode(v, links[v.name]) for v in vs], [FactorNode(f) for f in facs],
                      noise=0.01, rng=np.random.default_rng(7))
    assert g.init_idx is not None
    compare_with_oracle(oracle_built, g, Params(), 0, steps=[0, 1, 3, 12])
