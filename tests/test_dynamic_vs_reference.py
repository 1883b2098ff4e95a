"""SURVEY 8(f).3 pinned against the REFERENCE: `change_factor_function` with the same scope (also
in another dimension order) and the external-variable slice, on the reference's own computation
objects (oracle/ref_harness.ReferenceMaxSumRun: the reference's `DynamicFunctionFactorComputation.
change_factor_function`, maxsum_dynamic.py:80-104, and `relation.slice`, relations.py:760-810)
against the oracle and the (emulated) engine: `mxs_update_factor_table`, `mxs_set_parent_table` +
`mxs_slice_factor`.  Mirrors the reference's tests/unit/test_algorithms_dynamic_maxsum.py:61-105
(same scope, different dimension order, wrong dimensions -> ValueError).

What stays unpinned, and why (checked below): a change of SCOPE -- the reference's classes for it
(`DynamicFactorComputation`, `FactorWithReadOnlyVariableComputation`,
`DynamicFactorVariableComputation`) cannot be constructed at all."""
import numpy as np
import pytest

from oracle import ref_harness
from parity_common import assert_messages_equal_reference
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


def _relation(dcop, g, f, order, table):
    """An extensional relation over factor f's variables in scope positions `order`, `table`
    indexed in that order."""
    from pydcop.dcop.relations import NAryMatrixRelation
    scope = [int(v) for v in g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]]]
    variables = [dcop.variables[g.var_names[scope[i]]] for i in order]
    return NAryMatrixRelation(variables, np.asarray(table), name=g.factor_names[f])


def _engines(g, mode, params, oracle_built, emu):
    from pydcop_amd.dynamic import DynamicMaxSum
    from pydcop_amd.engine import MaxSumEngine
    p = Params(mode=mode, **params)
    ora = oracle_built.OracleMaxSum(g, p)
    run = DynamicMaxSum(g, p, engine_factory=lambda graph, pp: MaxSumEngine(graph, pp, lib_path=emu))
    return ora, run


@pytest.fixture(scope="module")
def emu():
    from emu.build_emu import build
    return build()


CASES = [("mixed_min", lambda: G.random_mixed(24, 36, seed=31, max_arity=3, dom_choices=(2, 3, 4)), "min", {}),
         ("mixed_max_lv", lambda: G.random_mixed(24, 36, seed=32, max_arity=3, dom_choices=(2, 3)), "max",
          {"start_messages": "leafs_vars", "damping_nodes": "vars"}),
         ("coloring_all", lambda: G.random_coloring(30, seed=33), "min", {"start_messages": "all"})]


@pytest.mark.parametrize("name,make,mode,params", CASES, ids=[c[0] for c in CASES])
def test_change_factor_function_same_scope_and_permuted(name, make, mode, params, oracle_built, emu):
    check_change_factor_function(make, mode, params, oracle_built, emu)


def check_change_factor_function(make, mode, params, oracle_built, emu):
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    ref = ref_harness.ReferenceMaxSumRun(dcop, params, cg=cg)
    ora, run = _engines(g, mode, params, oracle_built, emu)
    rng = np.random.default_rng(7)
    arity = np.diff(g.factor_rowptr)
    done = 0

    def advance(n, what):
        nonlocal done
        ora.run(n - 1), run.run(n - 1)
        before = ora.messages()
        ora.run(1), run.run(1)
        done += n
        ref.run_to(done)
        state = ref_harness.reference_message_state(ref.comps, g)
        assert_messages_equal_reference(state, ora.messages(), before)    # every message, every counter
        for x, y in zip(run.messages(), ora.messages()):
            np.testing.assert_array_equal(x, y, err_msg=what)
        vals, costs = ref.values()
        idx = np.array([g.domains[i].index(vals[n_]) for i, n_ in enumerate(g.var_names)])
        np.testing.assert_array_equal(ora.assignment()[0], idx, err_msg=what)
        np.testing.assert_array_equal(run.assignment()[0], idx, err_msg=what)
        np.testing.assert_allclose(ora.assignment()[1], [costs[n_] for n_ in g.var_names], rtol=1e-12, atol=1e-12)

    advance(4, "before any change")
    # same scope, same order: a new table for a binary and (where there is one) a ternary factor
    for ar in (2, 3):
        fs = np.flatnonzero(arity == ar)
        if not len(fs):
            continue
        f = int(fs[0])
        shape = [int(g.dom_size[v]) for v in g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]]]
        t = rng.integers(-6, 10, shape).astype(float) + (0.25 if ar == 3 else 0.0)
        ref.change_factor_function(g.factor_names[f], _relation(dcop, g, f, range(ar), t))
        ora.update_factor_table(f, t)
        run.change_factor_function(f, t)
        advance(3, f"same scope, arity {ar}")
    # the same variables in ANOTHER dimension order (tests/unit/test_algorithms_dynamic_maxsum.py:61-82)
    for ar in (2, 3):
        fs = np.flatnonzero(arity == ar)
        if len(fs) < 2:
            continue
        f = int(fs[1])
        scope = [int(v) for v in g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]]]
        order = list(range(ar))[::-1]
        t = rng.uniform(-3, 3, [int(g.dom_size[scope[i]]) for i in order])
        ref.change_factor_function(g.factor_names[f], _relation(dcop, g, f, order, t))
        ora.update_factor_table(f, np.ascontiguousarray(np.transpose(t, order)))   # back in the factor's own order
        run.change_factor_function(f, t, scope=[scope[i] for i in order])
        advance(3, f"permuted scope, arity {ar}")
    assert run.relayouts == 0   # same variables: no new layout, the messages carried on
    ora.close(), run.close()


def test_change_factor_function_rejects_other_dimensions():
    """maxsum_dynamic.py:88-99, tests/unit/test_algorithms_dynamic_maxsum.py:84-131."""
    g = G.random_coloring(12, seed=3)
    dcop, cg = ref_harness.flat_to_dcop(g, "min")
    ref = ref_harness.ReferenceMaxSumRun(dcop, cg=cg)
    from pydcop.dcop.relations import NAryMatrixRelation
    vs = [dcop.variables[n] for n in g.var_names]
    scope = [int(v) for v in g.edge_var[0:2]]
    other = [v for i, v in enumerate(vs) if i not in scope]
    with pytest.raises(ValueError):      # one variable more
        ref.change_factor_function(g.factor_names[0], NAryMatrixRelation([vs[scope[0]], vs[scope[1]], other[0]], name="x"))
    with pytest.raises(ValueError):      # same count, another variable
        ref.change_factor_function(g.factor_names[0], NAryMatrixRelation([vs[scope[0]], other[0]], name="x"))
    # the engine side of the same-scope entry point refuses a table of another size
    from emu.build_emu import build
    from pydcop_amd.engine import MaxSumEngine, MaxSumGpuError
    with MaxSumEngine(g, Params(), lib_path=build()) as eng:
        with pytest.raises((MaxSumGpuError, ValueError)):
            eng.update_factor_table(0, np.zeros(27))


def test_external_variable_slice_equals_reference_slice(oracle_built, emu):
    check_external_slice(oracle_built, emu)


def check_external_slice(oracle_built, emu):
    """`mxs_set_parent_table` + `mxs_slice_factor` == `relation.slice(values)` (what
    FactorWithReadOnlyVariableComputation computes, maxsum_dynamic.py:166) followed by
    change_factor_function(new_sliced): every message after every sensor move."""
    g = G.random_mixed(20, 30, seed=41, max_arity=2, dom_choices=(2, 3, 4))
    mode, params = "min", {"start_messages": "leafs_vars"}
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    ref = ref_harness.ReferenceMaxSumRun(dcop, params, cg=cg)
    ora, run = _engines(g, mode, params, oracle_built, emu)
    from pydcop.dcop.objects import Domain, Variable
    from pydcop.dcop.relations import NAryMatrixRelation
    f = int(np.flatnonzero(np.diff(g.factor_rowptr) == 2)[0])
    scope = [int(v) for v in g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]]]
    x, y = (dcop.variables[g.var_names[v]] for v in scope)
    sensor = Variable("sensor", Domain("s", "s", list(range(5))))
    rng = np.random.default_rng(5)
    parent = rng.integers(-9, 9, (len(x.domain), 5, len(y.domain))).astype(float)   # the sensor in the middle
    parent_rel = NAryMatrixRelation([x, sensor, y], parent, name=g.factor_names[f])
    run.register_external(f, parent, [0, 1, 0])
    done = 0
    for value in (3, 0, 4):
        sliced = ref.slice_external(parent_rel, {"sensor": value})
        assert [v.name for v in sliced.dimensions] == [x.name, y.name]
        np.testing.assert_array_equal(np.asarray(sliced._m), parent[:, value, :])
        ref.change_factor_function(g.factor_names[f], sliced)
        run.set_external_values(f, [value])
        ora.update_factor_table(f, np.ascontiguousarray(parent[:, value, :]))
        ora.run(2), run.run(2)
        before = ora.messages()
        ora.run(1), run.run(1)
        done += 3
        ref.run_to(done)
        assert_messages_equal_reference(ref_harness.reference_message_state(ref.comps, g), ora.messages(), before)
        for a, b in zip(run.messages(), ora.messages()):
            np.testing.assert_array_equal(a, b)
    ora.close(), run.close()


def test_which_reference_classes_cannot_be_constructed():
    """Why the scope change stays unpinned: the classes the reference has for it do not construct."""
    ref_harness.install_shims()
    import pydcop.algorithms.maxsum_dynamic as md
    from pydcop.dcop.objects import Variable
    from pydcop.dcop.relations import NAryMatrixRelation
    x, y = Variable("x", [0, 1]), Variable("y", [0, 1])
    rel = NAryMatrixRelation([x, y], np.zeros((2, 2)), name="r")
    with pytest.raises(TypeError):     # passes (relation, name=, msg_sender=) to a parent that takes comp_def
        md.FactorWithReadOnlyVariableComputation(rel, [y])
    with pytest.raises(TypeError):
        md.DynamicFactorComputation(rel, name="r")
    with pytest.raises(TypeError):
        md.DynamicFactorVariableComputation(x, ["r"])


def check_reference_dynamic_class_under_amaxsum(make, mode, params, emu_lib):
    """The reference's `change_factor_function` (maxsum_dynamic.py:80-104) where the reference defines
    it: on the ASYNCHRONOUS factor computation (the parent class of DynamicFunctionFactorComputation,
    maxsum_dynamic.py:35), FIFO delivery, called between two generations (same scope, then the same
    variables in another dimension order) -- against `mxs_amaxsum_update_factor_table` of the engine
    and the asynchronous oracle: values, costs and every held / last-sent message, bit for bit, after
    every change."""
    from oracle.amaxsum_oracle import OracleAMaxSum
    from pydcop_amd.amaxsum import AMaxSumEngine
    from test_amaxsum_oracle_vs_reference import _held
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    ref = ref_harness.ReferenceDynamicAMaxSumRun(dcop, params, cg=cg)
    p = Params(mode=mode, **params)
    ora, eng = OracleAMaxSum(g, p), AMaxSumEngine(g, p, lib_path=emu_lib)
    rng = np.random.default_rng(11)
    arity = np.diff(g.factor_rowptr)

    def advance(gens, what):
        n = ref.run(gens)
        assert ora.run(gens) + 0 >= 0 and eng.run(gens) >= 0
        assert ora.delivered == eng.delivered == n, what
        held = _held(ref.comps, g, None)
        for k, want in held.items():
            np.testing.assert_array_equal(ora.messages()[k], want, err_msg=f"{k} {what}")
            np.testing.assert_array_equal(eng.messages()[k], want, err_msg=f"{k} {what}")
        vals, costs = ref.values()
        idx = np.array([g.domains[i].index(vals[nm]) for i, nm in enumerate(g.var_names)])
        for e in (ora, eng):
            np.testing.assert_array_equal(e.assignment()[0], idx, err_msg=what)
            np.testing.assert_array_equal(e.assignment()[1], [0.0 if costs[nm] is None else costs[nm] for nm in g.var_names],
                                          err_msg=what)

    advance(3, "before any change")
    gens = 3
    for ar, order_of in ((2, lambda a: list(range(a))), (3, lambda a: list(range(a))),
                         (2, lambda a: list(range(a))[::-1]), (3, lambda a: [1, 2, 0])):
        fs = np.flatnonzero(arity == ar)
        if not len(fs):
            continue
        f = int(fs[gens % len(fs)])
        scope = [int(v) for v in g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]]]
        order = order_of(ar)
        t = rng.integers(-6, 10, [int(g.dom_size[scope[i]]) for i in order]).astype(float) + 0.5
        ref.change_factor_function(g.factor_names[f], _relation(dcop, g, f, order, t))
        own = np.ascontiguousarray(np.transpose(t, np.argsort(order)))   # back in the factor's own dimension order
        ora.update_factor_table(f, own), eng.update_factor_table(f, own)
        gens += 3
        advance(gens, f"arity {ar}, order {order}")
    ora.close(), eng.close()


ACASES = [("mixed_lv", lambda: G.random_mixed(20, 30, seed=51, max_arity=3, dom_choices=(2, 3, 4)), "min", {"start_messages": "leafs_vars"}),
          ("mixed_max_all", lambda: G.random_mixed(16, 24, seed=52, max_arity=3, dom_choices=(2, 3)), "max", {"start_messages": "all"}),
          ("coloring_lv", lambda: G.random_coloring(24, seed=53), "min", {"start_messages": "leafs_vars", "damping_nodes": "vars"})]


@pytest.mark.parametrize("name,make,mode,params", ACASES, ids=[c[0] for c in ACASES])
def test_reference_dynamic_class_under_amaxsum(name, make, mode, params, oracle_built, emu):
    check_reference_dynamic_class_under_amaxsum(make, mode, params, emu)


def test_the_reference_dynamic_class_itself_cannot_handle_a_message():
    """Why `change_factor_function` is exercised on the parent class's objects: a
    DynamicFunctionFactorComputation constructs, but its handler table lacks the parent's
    `max_sum` handler -- the first message raises KeyError (computations.py:509; the class
    docstring says so: "does not work since the refactoring", maxsum_dynamic.py:60)."""
    g = G.random_coloring(12, seed=3)
    dcop, cg = ref_harness.flat_to_dcop(g, "min")
    run = ref_harness.ReferenceDynamicAMaxSumRun(dcop, {"start_messages": "leafs_vars"}, cg=cg, dynamic_class=True)
    assert run.q          # the variables' start messages are waiting for the factors
    with pytest.raises(KeyError):
        run.run(1)
