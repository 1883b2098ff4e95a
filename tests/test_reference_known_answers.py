"""The known answers the reference's OWN unit tests pin for this path,
restated on the flat graph format and checked against the oracle -- on every machine, the
reference checkout is not needed (inputs and expected values are copied from the cited
tests; tests/test_oracle_vs_reference.py runs the live reference where it exists)."""
import ctypes as C

import numpy as np
import pytest

from pydcop_amd.graph import FlatGraph, Params


def _graph(dom_sizes, var_cost, factors):
    """factors: list of (scope tuple, table ndarray)"""
    edge_var, rowptr, toff, tabs = [], [0], [0], []
    for scope, tab in factors:
        edge_var += list(scope)
        rowptr.append(len(edge_var))
        tabs.append(np.asarray(tab, dtype=float).reshape(-1))
        toff.append(toff[-1] + tabs[-1].size)
    vr, ve = FlatGraph.var_side_from_edges(np.array(edge_var, dtype=np.int64), len(dom_sizes))
    return FlatGraph(dom_size=dom_sizes, var_cost=var_cost, factor_rowptr=rowptr, edge_var=edge_var,
                     table_off=toff, tables=np.concatenate(tabs) if tabs else np.zeros(0),
                     var_rowptr=vr, var_edges=ve).validate()


def test_factor_cost_at_start(oracle_built):
    """tests/unit/test_algorithms_maxsum.py:102-112 -- factor "10 if v1 == v2 else 0" over
    {R, G}^2, no message received yet: factor_costs_for_var(c1, v1, {}, "min") == {R: 0, G: 0}."""
    g = _graph([2, 2], np.zeros(4), [((0, 1), [[10, 0], [0, 10]])])
    o = oracle_built.OracleMaxSum(g, Params(mode="min", start_messages="all"))  # cycle 0 = on_start
    _, f2v, _, _ = o.messages()
    np.testing.assert_array_equal(f2v, [0, 0, 0, 0])


def test_select_value_with_cost_function(oracle_built):
    """tests/unit/test_algorithms_maxsum.py:115-127 -- a variable over [1, 2, 3] with cost
    (4 - v) / 10 and no factor: select_value == (3, 0.1); without costs the cost is 0."""
    g = _graph([3, 3], np.array([(4 - v) / 10 for v in (1, 2, 3)] + [0, 0, 0]), [])
    o = oracle_built.OracleMaxSum(g, Params(mode="min"))
    idx, belief = o.assignment()
    assert idx[0] == 2 and belief[0] == 0.1   # value 3
    assert belief[1] == 0 and idx[1] in (0, 1, 2)


@pytest.mark.parametrize("mode", ["min"])
def test_cost_for_one_variable(oracle_built, mode):
    """tests/unit/test_algorithms_amaxsum.py:77-123 -- unary factor x1 * 2 over 0..9: the
    message is the factor function itself, costs[0] == 0 and costs[5] == 10."""
    g = _graph([10], np.zeros(10), [((0,), 2.0 * np.arange(10))])
    o = oracle_built.OracleMaxSum(g, Params(mode=mode))  # unary factors send at start in every mode
    _, f2v, _, _ = o.messages()
    assert f2v[0] == 0 and f2v[5] == 10
    np.testing.assert_array_equal(f2v, 2.0 * np.arange(10))


def test_cost_for_two_variables(oracle_built):
    """tests/unit/test_algorithms_amaxsum.py:125-150 -- abs((x1 - x2) / 2), x1 in 0..9, x2 in
    0..4, nothing received: for x1 = 5 the best x2 is 4 (cost 0.5), x1 = 9 -> 2.5, x1 = 2 -> 0."""
    tab = np.abs((np.arange(10)[:, None] - np.arange(5)[None, :]) / 2)
    g = _graph([10, 5], np.zeros(15), [((0, 1), tab)])
    o = oracle_built.OracleMaxSum(g, Params(mode="min", start_messages="all"))
    _, f2v, _, _ = o.messages()
    to_x1 = f2v[:10]
    assert to_x1[5] == (5 - 4) / 2 and to_x1[9] == (9 - 4) / 2 and to_x1[2] == 0


def test_approx_match_cases(oracle_built):
    """tests/unit/test_algorithms_amaxsum.py:160-203 (ApproxMatchTests): identical zeros match;
    one differing component does not; a vector against all zeros does not (prev + c != 0 but the
    relative change is 2)."""
    from oracle.maxsum_oracle import _lib
    for dtype in ("f64", "f32"):
        lib = _lib(dtype)
        lib.mso_approx_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double]
        lib.mso_approx_match.restype = C.c_int

        def match(c1, c2):
            a, b = np.asarray(c1, dtype=np.float64), np.asarray(c2, dtype=np.float64)
            return lib.mso_approx_match(a.ctypes.data, b.ctypes.data, a.size, 0.1)

        assert match([0, 0, 0], [0, 0, 0]) == 1
        assert match([0, 0, 0], [0, 1, 0]) == 0
        c1 = [-46.0, -46.5, -55.5, -56.0, -56.5, -65.5, -66.0, -66.5, -67.0, -67.5]
        assert match(c1, [0.0] * 10) == 0
        assert match([100.0, 50.0], [101.0, 50.0]) == 1   # 2 * 1 / 201 < 0.1
        assert match([1.0, 50.0], [2.0, 50.0]) == 0       # 2 * 1 / 3 > 0.1
