"""pydcop_amd -- MI355X-native synchronous Max-Sum engine behind pyDCOP's
algorithm-plugin API (`pydcop solve --algo maxsum_gpu`).

Only the hot path of pydcop/algorithms/maxsum.py is implemented here, as
hand-written HIP kernels for gfx950 behind the C-ABI of include/maxsum_gpu.h.
There is no CPU fallback: importing works anywhere, creating an engine needs
the built `libmaxsum_hip.so` and an MI355X.
"""
from .graph import FlatGraph, Params  # noqa: F401

__version__ = "0.1.0"
