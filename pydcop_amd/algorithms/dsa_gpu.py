"""dsa_gpu -- the reference's DSA (pydcop/algorithms/dsa.py, variants A / B / C) on the GPU, behind
the algorithm-module contract (same GRAPH_TYPE `constraints_hypergraph`, the same four parameters
with the same defaults, the same footprint / load formulas), reusing the proxies and the session of
`maxsum_gpu`.

One cycle of DSA = one launch over all variables (pydcop_amd/csrc/dsa.hip).  `stop_cycle: n` ends
after n cycles like the reference (dsa.py:351-354); 0 = keep going, `chunk` cycles per report, until
the orchestrator's timeout.  Extra parameter `seed` (default 0): every stochastic choice -- initial
value, move test, choice among equally good values -- comes from a counter-based generator keyed on
(seed, variable, cycle, draw), where the reference draws from Python's unseeded `random`: a run is
reproducible, and bit for bit the reference's own DsaComputation under the same generator.
"""
from pydcop.algorithms import AlgoParameterDef

from pydcop_amd.algorithms import maxsum_gpu as _base
from pydcop_amd.algorithms.mgm_gpu import _MgmSession
from pydcop_amd.graph import Params

GRAPH_TYPE = "constraints_hypergraph"
HEADER_SIZE = 0
UNIT_SIZE = 1

algo_params = [
    AlgoParameterDef("probability", "float", None, 0.7),
    AlgoParameterDef("p_mode", "str", ["fixed", "arity"], "fixed"),
    AlgoParameterDef("variant", "str", ["A", "B", "C"], "B"),
    AlgoParameterDef("stop_cycle", "int", None, 0),
    AlgoParameterDef("precision", "str", ["f64", "f32"], "f64"),
    AlgoParameterDef("seed", "int", None, 0),
    AlgoParameterDef("chunk", "int", None, 10),
]


def computation_memory(computation) -> float:
    """pydcop/algorithms/dsa.py:138-158: one value per neighbour."""
    neighbors = set(n for link in computation.links for n in link.nodes if n not in computation.name)
    return len(neighbors) * UNIT_SIZE


def communication_load(src, target: str) -> float:
    """pydcop/algorithms/dsa.py:161-183."""
    return UNIT_SIZE + HEADER_SIZE


class _CycleEngine:
    """DsaEngine behind the surface the session drives."""

    def __init__(self, graph, params, p):
        from pydcop_amd.dsa import DsaEngine
        self.graph = graph
        self._e = DsaEngine(graph, params, variant=p["variant"], probability=float(p["probability"]),
                            p_mode=p["p_mode"], seed=int(p["seed"]))

    def run(self, n: int):
        self._e.run(int(n))

    def assignment(self):
        return self._e.assignment()

    @property
    def cycle_count(self) -> int:
        return self._e.cycle_count

    def close(self):
        self._e.close()


class _DsaSession(_MgmSession):      # (same compilation of the hypergraph nodes as mgm_gpu)
    ALGO = "dsa_gpu"

    def _engine_params(self, algo, p):
        return Params(mode=algo.mode, dtype=p["precision"])

    def _make_engine(self, params, p):
        return _CycleEngine(self.graph, params, p)

    def _cycles_to_run(self, p) -> int:
        return int(p["stop_cycle"])

    def update_factor(self, name, old, fn):
        raise ValueError("dsa_gpu: change_factor_function is a maxsum_gpu feature")


_base.SESSION_CLASSES["dsa_gpu"] = _DsaSession


class DsaGpuComputation(_base.MaxSumGpuVariableComputation):
    """Stands for a DsaComputation (pydcop/algorithms/dsa.py:214)."""

    def footprint(self) -> float:
        return computation_memory(self.computation_def.node)


def build_computation(comp_def):
    if comp_def.node.type != "VariableComputationNode":
        raise ValueError("dsa_gpu: unsupported computation node type " + str(comp_def.node.type))
    return DsaGpuComputation(comp_def)
