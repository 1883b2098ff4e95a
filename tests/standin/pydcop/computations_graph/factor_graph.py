"""TEST stand-in for the node classes of pydcop/computations_graph/factor_graph.py:45-207:
only the attributes the plugin and pydcop_amd.compile read."""


class FactorGraphLink:
    def __init__(self, factor_node, variable_node):
        self.factor_node, self.variable_node = factor_node, variable_node
        self.type = "factor_graph_link"


class VariableComputationNode:
    type = "VariableComputation"

    def __init__(self, variable, constraints_names):
        self.variable, self.name = variable, variable.name
        self.constraints_names = list(constraints_names)
        self.links = [FactorGraphLink(c, self.name) for c in self.constraints_names]
        self.neighbors = list(self.constraints_names)


class FactorComputationNode:
    type = "FactorComputation"

    def __init__(self, factor):
        self.factor, self.name = factor, factor.name
        self.variables = list(factor.dimensions)
        self.links = [FactorGraphLink(self.name, v.name) for v in self.variables]
        self.neighbors = [v.name for v in self.variables]
