"""generators.peav_like against the reference's OWN meeting-scheduling model (PEAV):
pydcop/commands/generators/meetingscheduling.py `peav_model` is fed the problem definition our generator
drew (resources, events) and must produce the same variables (names order, domains) and, constraint by
constraint, the same scope and the same table -- entry for entry, bit for bit (the utilities are
`1 / (n - 1) * (v1 + v2)` in both).  Where /root/reference (or the staged archive) is not on the
machine the test is skipped."""
import numpy as np
import pytest

from oracle import ref_harness as RH
from pydcop_amd import generators as G

pytestmark = pytest.mark.skipif(not RH.reference_available(), reason="reference checkout not on this machine")


@pytest.mark.parametrize("kw", [dict(n_events=9, n_resources=5, slots=6, max_length=3, max_resources_event=3, seed=1),
                                dict(n_events=14, n_resources=6, slots=9, max_length=4, max_resources_event=4, seed=2),
                                dict(n_events=6, n_resources=7, slots=23, max_length=7, max_resources_event=5, seed=3)])
def test_peav_like_is_the_reference_model(kw):
    RH.install_shims()
    from pydcop.commands.generators import meetingscheduling as M
    max_value = 10
    value_free, lengths, events = G.peav_problem(kw["n_events"], kw["n_resources"], kw["slots"], kw["max_length"],
                                                 kw["max_resources_event"], max_value, kw["seed"])
    slots = list(range(1, kw["slots"] + 1))
    ref_resources = {r: M.Resource(r, {t: int(value_free[r, t]) for t in slots}) for r in range(kw["n_resources"])}
    ref_events = {e: M.Event(e, {int(r): int(v) for r, v in zip(*events[e])}, int(lengths[e])) for e in range(kw["n_events"])}
    penalty = max_value * kw["slots"] * kw["n_resources"]                # meetingscheduling.py:222
    variables, constraints, _ = M.peav_model(slots, ref_events, ref_resources, penalty)
    g = G.peav_like(**kw, max_value=max_value, unary_noise=0.0)
    # variables: the reference iterates resources, then events -- ours in the same order, same domains
    ref_vars = list(variables.values())
    assert len(ref_vars) == g.n_vars
    index = {v.name: i for i, v in enumerate(ref_vars)}
    for i, v in enumerate(ref_vars):
        assert list(v.domain.values) == list(range(int(g.dom_size[i]))), v.name
    # constraints: as multisets of (scope, table) -- the reference keys them by name, we by position
    def key(scope, table):
        return (tuple(scope), np.ascontiguousarray(table, dtype=np.float64).tobytes())
    ref_set = sorted(key([index[v.name] for v in c.dimensions], c._m) for c in constraints.values())
    ours = sorted(key(g.edge_var[g.factor_rowptr[f]:g.factor_rowptr[f + 1]].tolist(),
                      g.tables[g.table_off[f]:g.table_off[f + 1]]) for f in range(g.n_factors))
    assert len(ref_set) == len(ours) == g.n_factors
    assert ref_set == ours
