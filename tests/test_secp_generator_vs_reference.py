"""generators.secp_like against the reference's OWN expressions (pydcop/commands/generators/secp.py):
the reference draws its parameters from Python's unseeded `random`, so the test goes the other way --
the parameters OUR generator drew (efficiencies, impacts, targets, scopes) are written as the
expression strings build_lights / build_models / build_rules write (:224-226, :279-291, :309-313), handed to
the reference's `constraint_from_str`, and every table entry the reference computes must equal ours bit
for bit.  Skipped where the reference is not on the machine."""
import itertools

import numpy as np
import pytest

from oracle import ref_harness as RH
from pydcop_amd import generators as G

pytestmark = pytest.mark.skipif(not RH.reference_available(), reason="reference checkout not on this machine")


@pytest.mark.parametrize("kw", [dict(n_lights=12, n_models=7, n_rules=14, max_model_size=3, max_rule_size=3, seed=1),
                                dict(n_lights=9, n_models=6, n_rules=10, max_model_size=4, max_rule_size=3, seed=2)])
def test_secp_like_tables_are_the_reference_expressions(kw):
    RH.install_shims()
    from pydcop.dcop.objects import Domain, Variable
    from pydcop.dcop.relations import constraint_from_str
    g, spec = G.secp_like(**kw, names=False, return_spec=True)
    nl, nm = kw["n_lights"], kw["n_models"]
    dom = Domain("light", "light", range(0, 5))                                   # secp.py:138
    name = lambda v: f"l{v}" if v < nl else f"m{v - nl}"                          # noqa: E731
    variables = {v: Variable(name(v), dom) for v in range(nl + nm)}
    all_vars = list(variables.values())

    def ref_table(expr, scope):
        c = constraint_from_str("c", expr, all_vars)
        assert sorted(v.name for v in c.dimensions) == sorted(name(v) for v in scope)
        out = np.empty((5,) * len(scope))
        for vals in itertools.product(range(5), repeat=len(scope)):
            out[vals] = c(**{name(v): x for v, x in zip(scope, vals)})
        return out.reshape(-1)

    tables = {}          # scope tuple -> list of tables (ours), in factor order
    for f in range(g.n_factors):
        sc = tuple(g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]].tolist())
        tables.setdefault(sc, []).append(g.tables[g.table_off[f]:g.table_off[f + 1]])
    checked = 0

    def check(scope, expr):
        nonlocal checked
        want = ref_table(expr, scope)
        assert any(np.array_equal(want, t) for t in tables[tuple(scope)]), (scope, expr)
        checked += 1

    for i, e in enumerate(spec["efficiency"]):                                    # build_lights :309-313
        check([i], "{} * {}".format(name(i), float(e)))
    for scopes, impacts in spec["models"]:                                        # build_models :216-227
        for sc, imp in zip(scopes.tolist(), impacts.tolist()):
            lights, model = sc[:-1], sc[-1]
            light_expression = " + ".join(" {} * {}".format(name(v), w) for v, w in zip(lights, imp))
            check(sc, f"0 if 10* abs({name(model)} - ({light_expression})) < 5 else 10000 ")
    for scopes, targets in spec["rules"]:                                         # build_rules :271-291
        for sc, tg in zip(scopes.tolist(), targets.tolist()):
            expression = " + ".join(f"abs({name(v)} - {int(t)} )" for v, t in zip(sc, tg))
            check(sc, f"10 * ({expression})")
    assert checked == g.n_factors
    assert g.n_vars == nl + nm and (g.dom_size == 5).all()
