"""pyDCOP algorithm plugins shipped by pydcop_amd.

`pydcop_amd.plugin.install()` appends this directory to
`pydcop.algorithms.__path__`, which makes `maxsum_gpu` discoverable by
`list_available_algorithms` / `load_algorithm_module`
(pydcop/algorithms/__init__.py:508-566) with the reference untouched.
"""
