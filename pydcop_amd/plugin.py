"""Make `maxsum_gpu` visible to an unmodified pyDCOP and run its CLI.

    python -m pydcop_amd.plugin -t 5 solve --algo maxsum_gpu -p stop_cycle:30 \
           -d adhoc tests/instances/graph_coloring1.yaml

is `pydcop -t 5 solve ...` (pydcop/dcop_cli.py:62) with this package's
algorithm directory appended to `pydcop.algorithms.__path__`, which is where
`list_available_algorithms` / `load_algorithm_module`
(pydcop/algorithms/__init__.py:508-566) look for plugins -- verified in
SURVEY.md section 8b.  pyDCOP itself must be importable (PYTHONPATH).
"""
import collections
import collections.abc
import os
import sys
import types


def _compat_shims():
    """pyDCOP v0.1.2a1 predates python 3.10 / numpy 2 and imports two optional
    third-party packages at module level; make it importable as-is."""
    for n in ("Iterable", "Mapping", "Sequence", "Callable", "Sized", "MutableMapping",
              "Hashable", "Set"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))

    class _Permissive(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            stub = type(name, (), {"__init__": lambda self, *a, **k: None})
            setattr(self, name, stub)
            return stub

    for mod in ("websocket_server", "websocket_server.websocket_server",
                "pulp", "pulp.constants", "pulp.pulp", "pulp.solvers"):
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                sys.modules[mod] = _Permissive(mod)


def install(fast_graph=None):
    """Append pydcop_amd/algorithms to pydcop.algorithms.__path__ and
    pydcop_amd/computations_graph to pydcop.computations_graph.__path__ (idempotent).

    fast_graph=True (or $MAXSUM_GPU_GRAPH=fast) makes `maxsum_gpu` use the O(E)
    builder `factor_graph_fast` instead of the reference's O(V*F)
    `factor_graph.build_computation_graph`; the graph it returns is the same."""
    _compat_shims()
    import pydcop.algorithms as algos
    import pydcop.computations_graph as graphs
    base = os.path.dirname(os.path.abspath(__file__))
    here = os.path.join(base, "algorithms")
    if here not in list(algos.__path__):
        algos.__path__.append(here)
    gdir = os.path.join(base, "computations_graph")
    if gdir not in list(graphs.__path__):
        graphs.__path__.append(gdir)
    if fast_graph is None:
        fast_graph = os.environ.get("MAXSUM_GPU_GRAPH", "") == "fast"
    if fast_graph:
        from pydcop.algorithms import load_algorithm_module
        load_algorithm_module("maxsum_gpu").GRAPH_TYPE = "factor_graph_fast"
    return here


def main(argv=None):
    install()
    from pydcop import dcop_cli
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    return dcop_cli.main()


if __name__ == "__main__":
    main()
