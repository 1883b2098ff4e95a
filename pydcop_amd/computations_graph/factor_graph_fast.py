"""factor_graph_fast -- the reference's factor graph, built in O(E).

`pydcop.computations_graph.factor_graph.build_computation_graph`
(factor_graph.py:245-296) scans every constraint for every variable
(`find_dependent_relations`, pydcop/dcop/relations.py:1219) -- O(V*F): 184 s for a
10 000-variable instance -- and `ComputationGraph.computation`
/ `links_for_node` / `neighbors` (computations_graph/objects.py:245-320) are linear
scans that the orchestrator calls once per deployed computation (O(N^2)).

This module produces the SAME graph -- the reference's own node and link classes, the
same node order (variables then factors, each in the DCOP's order) and the same
`links` order of every variable (its constraints in the order of the constraint list,
which is what `find_dependent_relations` returns) -- with one pass over the constraint
scopes, and name-indexed lookups.  Drop-in: set an algorithm module's
`GRAPH_TYPE = "factor_graph_fast"` (see pydcop_amd.plugin.install(fast_graph=True)).
"""
from collections import OrderedDict
from typing import Iterable

from pydcop.computations_graph.factor_graph import (ComputationsFactorGraph,  # noqa: F401
                                                     FactorComputationNode, FactorGraphLink,
                                                     VariableComputationNode)
from pydcop.dcop.dcop import DCOP
from pydcop.dcop.objects import Variable
from pydcop.dcop.relations import Constraint

GRAPH_NODE_TYPES = ("VariableComputation", "FactorComputation")


class FastComputationsFactorGraph(ComputationsFactorGraph):
    """A ComputationsFactorGraph with dictionary lookups by node name."""

    def __init__(self, var_nodes: Iterable[VariableComputationNode],
                 factor_nodes: Iterable[FactorComputationNode]) -> None:
        super().__init__(var_nodes, factor_nodes)  # checks for duplicate names
        self._by_name = {n.name: n for n in self.nodes}

    def computation(self, node_name: str):
        try:
            return self._by_name[node_name]
        except KeyError:
            raise KeyError("no computation named {} found".format(node_name))

    def links_for_node(self, node_name: str):
        try:
            return self._by_name[node_name].links
        except KeyError:
            raise KeyError("No node named " + node_name)

    def neighbors(self, node_name: str):
        try:
            return self._by_name[node_name].neighbors
        except KeyError:
            raise KeyError("No node named " + node_name)

    def node_names(self):
        return list(self._by_name)


def build_computation_graph(dcop: DCOP = None, variables: Iterable[Variable] = None,
                            constraints: Iterable[Constraint] = None) -> FastComputationsFactorGraph:
    """Same contract as factor_graph.build_computation_graph (factor_graph.py:245-296)."""
    if dcop is not None:
        if constraints or variables is not None:
            raise ValueError("Cannot use both dcop and constraints / variables parameters")
        variables = dcop.variables.values()
        constraints = dcop.constraints.values()
    elif constraints is None or variables is None:
        raise ValueError("Constraints AND variables parameters must be provided when not "
                         "building the graph from a dcop")
    variables, constraints = list(variables), list(constraints)
    depends = OrderedDict((v.name, []) for v in variables)
    for c in constraints:  # one pass over the scopes; keeps the constraint order per variable
        for v in c.dimensions:
            lst = depends.get(v.name)
            if lst is not None:
                lst.append(c.name)
    var_nodes = [VariableComputationNode(v, constraints_names=depends[v.name]) for v in variables]
    factor_nodes = [FactorComputationNode(c) for c in constraints]
    return FastComputationsFactorGraph(var_nodes, factor_nodes)
