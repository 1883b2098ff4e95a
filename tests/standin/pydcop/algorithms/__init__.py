"""TEST stand-in for the names maxsum_gpu imports from pydcop.algorithms
(the real ones: pydcop/algorithms/__init__.py:85-135 AlgoParameterDef, :141-290 AlgorithmDef,
:336-380 ComputationDef)."""
from collections import namedtuple

AlgoParameterDef = namedtuple("AlgoParameterDef", ["name", "type", "values", "default_value"])
__path__ = list(__path__)  # the launcher appends the plugin directory here, like on the real package

_CAST = {"int": int, "float": float, "str": str}


class AlgorithmDef:
    def __init__(self, algo, params, mode="min"):
        self.algo, self.params, self.mode = algo, dict(params), mode

    @staticmethod
    def build_with_default_param(algo, params=None, mode="min", parameters_definitions=None):
        if parameters_definitions is None:
            parameters_definitions = load_algorithm_module(algo).algo_params
        given = dict(params or {})
        out = {}
        for d in parameters_definitions:
            v = given.pop(d.name, d.default_value)
            if v is not None:
                v = _CAST[d.type](v)
                if d.values and v not in d.values:
                    raise ValueError(f"{d.name}: {v!r} not in {d.values}")
            out[d.name] = v
        if given:
            raise ValueError(f"unknown parameter(s) {sorted(given)} for {algo}")
        return AlgorithmDef(algo, out, mode)


class ComputationDef:
    def __init__(self, node, algo):
        self.node, self.algo = node, algo

    @property
    def name(self):
        return self.node.name


def load_algorithm_module(name):
    from importlib import import_module
    return import_module("pydcop.algorithms." + name)
