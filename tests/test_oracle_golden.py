"""The CPU oracle (oracle/maxsum_oracle.c) against the golden vectors that
oracle/make_golden.py generated from the reference's own computations
(pydcop/algorithms/maxsum.py) -- runs everywhere, no reference needed."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from parity_common import assert_messages_equal_reference
from pydcop_amd.graph import Params


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_reference_golden(path, oracle_built):
    g, params, meta, ref_idx, ref_cost = load_golden(path)
    o = oracle_built.OracleMaxSum(g, Params(**params))
    o.run(meta["T"] - 1)
    before = o.messages()
    o.run(1)
    # every message the reference's computations sent / hold + every send counter, bit for bit
    assert_messages_equal_reference(meta["ref_messages"], o.messages(), before)
    idx, belief = o.assignment()
    assert o.cycle_count == meta["T"]
    np.testing.assert_array_equal(idx, ref_idx)
    ok = ~np.isnan(ref_cost)
    # tolerance: select_value sums in dict-arrival order in the reference
    # (maxsum.py:609), in links order here -> last-bit differences only
    np.testing.assert_allclose(belief[ok], ref_cost[ok], rtol=1e-12, atol=1e-12)
    cost, viol = o.eval_cost(ref_idx)
    assert viol == meta["violation"]
    assert cost == pytest.approx(meta["cost"], rel=1e-12, abs=1e-9)


def test_golden_known_answers():
    """The answers the reference's own tests pin for this path
    (tests/dcop_cli/test_solve.py:39-72,100-130; instance header of
    tests/instances/graph_coloring_tuto.yaml:6-7)."""
    import glob
    from conftest import GOLDEN_DIR
    want = {
        "yaml_graph_coloring1_T20": {"v1": "R", "v2": "G", "v3": "R"},
        "yaml_secp_simple1_T20": {"l1": "0", "l2": "3", "l3": "4", "m1": "3"},
        "yaml_graph_coloring_tuto_T20": {"v1": "G", "v2": "G", "v3": "G", "v4": "G"},
    }
    for name, expect in want.items():
        _, _, meta, _, _ = load_golden(os.path.join(GOLDEN_DIR, name + ".npz"))
        got = dict(zip(meta["var_names"], meta["values"]))
        assert got == expect
    _, _, meta, _, _ = load_golden(os.path.join(GOLDEN_DIR, "yaml_graph_coloring_tuto_T20.npz"))
    assert meta["cost"] == 12


def test_oracle_f32_twin_close_to_f64(oracle_built):
    from pydcop_amd.generators import random_coloring
    g = random_coloring(300, seed=5)
    a = oracle_built.OracleMaxSum(g, Params(dtype="f64"))
    b = oracle_built.OracleMaxSum(g, Params(dtype="f32"))
    a.run(10), b.run(10)
    ia, ba = a.assignment()
    ib, bb = b.assignment()
    assert (ia != ib).mean() < 0.02
    np.testing.assert_allclose(ba[ia == ib], bb[ia == ib], rtol=1e-3, atol=1e-3)
