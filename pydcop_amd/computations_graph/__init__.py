"""Computation-graph builders shipped by pydcop_amd.

`pydcop_amd.plugin.install()` appends this directory to
`pydcop.computations_graph.__path__`, so that `GRAPH_TYPE = "factor_graph_fast"`
resolves through the reference's own `import_module("pydcop.computations_graph." +
GRAPH_TYPE)` (pydcop/commands/_utils.py:203-205, pydcop/infrastructure/run.py:109-111).
"""
