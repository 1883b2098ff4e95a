"""mgm_gpu -- the reference's MGM (pydcop/algorithms/mgm.py) on the GPU, behind the algorithm-module
contract (same GRAPH_TYPE `constraints_hypergraph`, the same two parameters with the same defaults,
the same footprint / load formulas), reusing the proxies and the session of `maxsum_gpu`.

One round of MGM = two launches over all variables (pydcop_amd/csrc/mgm.hip).  `stop_cycle: n`
ends like the reference does: after n - 1 rounds (its cycle counter starts at 1, mgm.py:407-411);
0 = keep going, `chunk` rounds per report, until the orchestrator's timeout.  The reference's draws
from the unseeded `random` module are fixed: first domain value at start (unless the variable has
an initial value), first of equally good values; ties between equal gains by name (the
reference's `break_mode: random` never triggers -- mgm.py:543 compares the string with the module
-- so both modes are lexic, here as there).
"""
from types import SimpleNamespace

from pydcop.algorithms import AlgoParameterDef

from pydcop_amd.algorithms import maxsum_gpu as _base
from pydcop_amd.compile import compile_nodes
from pydcop_amd.graph import Params

GRAPH_TYPE = "constraints_hypergraph"
HEADER_SIZE = 100
UNIT_SIZE = 5

algo_params = [
    AlgoParameterDef("break_mode", "str", ["lexic", "random"], "lexic"),
    AlgoParameterDef("stop_cycle", "int", None, 0),
    AlgoParameterDef("precision", "str", ["f64", "f32"], "f64"),
    AlgoParameterDef("chunk", "int", None, 10),
]


def computation_memory(computation) -> float:
    """pydcop/algorithms/mgm.py:82-112: one value per neighbour."""
    neighbors = set(n for link in computation.links for n in link.nodes if n not in computation.name)
    return len(neighbors) * UNIT_SIZE


def communication_load(src, target: str) -> float:
    """pydcop/algorithms/mgm.py:115-135."""
    return UNIT_SIZE + HEADER_SIZE


class _RoundEngine:
    """MgmEngine behind the surface the session drives."""

    def __init__(self, graph, params):
        from pydcop_amd.mgm import MgmEngine
        self.graph = graph
        self._e = MgmEngine(graph, params)

    def run(self, n: int):
        self._e.run(int(n))

    def assignment(self):
        return self._e.assignment()

    @property
    def cycle_count(self) -> int:
        return self._e.cycle_count + 1          # the reference's counter starts at 1 (mgm.py:407)

    def close(self):
        self._e.close()


class _MgmSession(_base._Session):
    ALGO = "mgm_gpu"

    def _compile_graph(self, p):
        nodes = sorted((cd.node for cd in self.comp_defs.values()), key=lambda n: n.name)
        constraints = {}
        for n in nodes:
            for c in n.constraints:
                constraints.setdefault(c.name, c)
        fac_nodes = [SimpleNamespace(name=name, factor=constraints[name]) for name in sorted(constraints)]
        var_nodes = [SimpleNamespace(name=n.name, variable=n.variable,
                                     links=[SimpleNamespace(factor_node=c.name) for c in n.constraints])
                     for n in nodes]
        return compile_nodes(var_nodes, fac_nodes, noise=0.0)

    def _engine_params(self, algo, p):
        return Params(mode=algo.mode, dtype=p["precision"])

    def _make_engine(self, params, p):
        return _RoundEngine(self.graph, params)

    def _cycles_to_run(self, p) -> int:
        stop = int(p["stop_cycle"])
        if stop == 1:          # the reference finishes before the first exchange of values
            self.done = True
        return max(0, stop - 1)

    def update_factor(self, name, old, fn):
        raise ValueError("mgm_gpu: change_factor_function is a maxsum_gpu feature")


_base.SESSION_CLASSES["mgm_gpu"] = _MgmSession


class MgmGpuComputation(_base.MaxSumGpuVariableComputation):
    """Stands for an MgmComputation (pydcop/algorithms/mgm.py:213)."""

    def footprint(self) -> float:
        return computation_memory(self.computation_def.node)


def build_computation(comp_def):
    if comp_def.node.type != "VariableComputationNode":
        raise ValueError("mgm_gpu: unsupported computation node type " + str(comp_def.node.type))
    return MgmGpuComputation(comp_def)
