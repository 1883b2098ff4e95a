"""amaxsum_gpu -- the reference's ASYNCHRONOUS Max-Sum (pydcop/algorithms/amaxsum.py) on the GPU,
behind the same algorithm-module contract as `maxsum_gpu` (whose proxies and session it reuses).

The reference's amaxsum handles one message at a time and what it computes depends on the order
the agents' threads deliver them in.  This module runs it under the one order that is defined
without a scheduler -- every computation started in graph order, one first-in-first-out queue --
a GENERATION of messages per device step (pydcop_amd/csrc/amaxsum.hip; bit for bit the
reference's own computations under that order -- pinned in tests/).

Parameters: those of amaxsum (= maxsum's, amaxsum.py:105) plus
  stop_cycle  int, default 0   > 0: deliver exactly that many generations, then finished();
                               0: keep going, `chunk` generations per report, until no message
                               is left (the send rule of amaxsum.py:222-244 ends the run by itself)
  precision, seed, chunk       as maxsum_gpu
NB with the reference's default `start_messages: leafs` only leaf variables / unary factors speak
first (amaxsum.py:140-160, 307-318): on a graph without leaves nothing ever happens, here as there.
"""
from pydcop.algorithms import AlgoParameterDef

from pydcop_amd.algorithms import maxsum_gpu as _base
from pydcop_amd.algorithms.maxsum_gpu import (MaxSumGpuFactorComputation, MaxSumGpuVariableComputation,  # noqa: F401
                                               communication_load, computation_memory)

GRAPH_TYPE = "factor_graph"

algo_params = [p for p in _base.algo_params if p.name != "devices"]


class _GenerationEngine:
    """AMaxSumEngine behind the surface the session drives (run(n) = n more generations)."""

    def __init__(self, graph, params):
        from pydcop_amd.amaxsum import AMaxSumEngine
        self.graph = graph
        self._e = AMaxSumEngine(graph, params)

    def run(self, n: int):
        self._e.run(self._e.generation + 1 + int(n))

    def assignment(self):
        return self._e.assignment()

    @property
    def cycle_count(self) -> int:
        return self._e.generation + 1

    @property
    def quiescent(self) -> bool:
        return self._e.pending == 0

    def update_factor_table(self, factor: int, table):
        self._e.update_factor_table(factor, table)

    def close(self):
        self._e.close()


class _AsyncSession(_base._Session):
    ALGO = "amaxsum_gpu"

    def _make_engine(self, params, p):
        return _GenerationEngine(self.graph, params)

    # change_factor_function: the base session's path for engines without re-layout -- the same
    # variables (any dimension order) swap the table in place between two generations
    # (mxs_amaxsum_update_factor_table; DynamicFunctionFactorComputation.change_factor_function,
    # maxsum_dynamic.py:80-104, allows nothing else); other variables raise ValueError there too.


_base.SESSION_CLASSES["amaxsum_gpu"] = _AsyncSession


def build_computation(comp_def):
    return _base.build_computation(comp_def)
