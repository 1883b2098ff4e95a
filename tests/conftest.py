import glob
import json
import os
import sys

import numpy as np
import pytest

try:
    # One HIP runtime per process: importing torch first makes the engine bind to
    # the runtime torch bundles (pydcop_amd/engine.py:hip_runtime_path), which the
    # GPU tests of the sharded path need (they hand device buffers to torch).
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# The CPU tests load the host emulation of the engine sources (tests/emu) where the product
# loads libmaxsum_hip.so; the product refuses such a library unless a test registers it.
from pydcop_amd import engine as _engine  # noqa: E402
_engine.register_test_engine(os.path.join(ROOT, "tests", "emu", "_build", "libmaxsum_emu.so"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(path):
    """-> (FlatGraph, Params kwargs, meta dict, ref_idx, ref_cost)"""
    from pydcop_amd.graph import FlatGraph
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    g = FlatGraph(dom_size=z["dom_size"], var_cost=z["var_cost"],
                  factor_rowptr=z["factor_rowptr"], edge_var=z["edge_var"],
                  table_off=z["table_off"], tables=z["tables"],
                  var_rowptr=z["var_rowptr"], var_edges=z["var_edges"],
                  init_idx=z["init_idx"] if "init_idx" in z.files else None)
    g.var_names = meta["var_names"]
    g.domains = meta["domains"]
    params = dict(mode=meta["mode"])
    params.update(meta["params"])
    keys = ("sent_f2v", "count_f2v", "sent_v2f", "count_v2f", "held_v2f", "held_f2v")
    if all("ref_" + k in z.files for k in keys):   # what the reference's computations hold (make_golden.py)
        meta["ref_messages"] = {k: z["ref_" + k] for k in keys}
    return g.validate(), params, meta, z["ref_idx"], z["ref_cost"]


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import maxsum_oracle
    maxsum_oracle.build()
    return maxsum_oracle
