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


def install():
    """Append pydcop_amd/algorithms to pydcop.algorithms.__path__ (idempotent)."""
    _compat_shims()
    import pydcop.algorithms as algos
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "algorithms")
    if here not in list(algos.__path__):
        algos.__path__.append(here)
    return here


def main(argv=None):
    install()
    from pydcop import dcop_cli
    if argv is not None:
        sys.argv = [sys.argv[0]] + list(argv)
    return dcop_cli.main()


if __name__ == "__main__":
    main()
