"""Shared checks: an engine (HIP on the GPU, or the emulated build on the CPU)
against the oracle and the golden vectors."""
import numpy as np

from pydcop_amd import generators as G
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params


def compare_with_oracle(oracle_mod, graph, params: Params, T, lib_path=None, exact=True,
                        steps=None, threads=1, expect_silent=False):
    """Run engine and oracle side by side; compare messages, counters, selection,
    beliefs and solution cost after every chunk of `steps` cycles."""
    eng = MaxSumEngine(graph, params, lib_path=lib_path)
    ora = oracle_mod.OracleMaxSum(graph, params, threads=threads)
    done = 0
    for n in (steps or [T]):
        eng.run(n)
        ora.run(n)
        done += n
        assert eng.cycle_count == ora.cycle_count == done
        ie, be = eng.assignment()
        io, bo = ora.assignment()
        me, mo = eng.messages(), ora.messages()
        if exact:
            np.testing.assert_array_equal(me[2], mo[2], err_msg=f"V->F counters, cycle {done}")
            np.testing.assert_array_equal(me[3], mo[3], err_msg=f"F->V counters, cycle {done}")
            np.testing.assert_array_equal(me[0], mo[0], err_msg=f"V->F messages, cycle {done}")
            np.testing.assert_array_equal(me[1], mo[1], err_msg=f"F->V messages, cycle {done}")
            np.testing.assert_array_equal(ie, io, err_msg=f"selection, cycle {done}")
            np.testing.assert_array_equal(be, bo, err_msg=f"beliefs, cycle {done}")
        else:
            np.testing.assert_allclose(me[0], mo[0], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(me[1], mo[1], rtol=1e-9, atol=1e-9)
            np.testing.assert_array_equal(ie, io)
            np.testing.assert_allclose(be, bo, rtol=1e-9, atol=1e-9)
        ce, ve = eng.eval_cost()
        co, vo = ora.eval_cost()
        assert ve == vo
        assert ce == co or (np.isnan(ce) and np.isnan(co)) or abs(ce - co) <= 1e-9 * max(1.0, abs(co))
    if expect_silent:  # the run must have reached the regime where edges stay silent (maxsum.py:371-377)
        assert (mo[2] == 4).any() or (mo[3] == 4).any(), "no send counter reached SAME_COUNT"
    eng.close()
    ora.close()


def assert_messages_equal_reference(ref, at_T, before_T):
    """`ref`: oracle/ref_harness.reference_message_state after T calls of on_new_cycle (live, or
    stored in a golden fixture); `at_T` / `before_T`: (v2f, f2v, count_v2f, count_f2v) of the
    flat engine (oracle or HIP) after T and T - 1 cycles.  Bit-exact: every expression of
    factor_costs_for_var / costs_for_factor / apply_damping runs in a deterministic order in the
    reference (dimensions / factors order).

    * what a sender last SENT + its counter (`_prev_messages`, maxsum.py:303, 474) == the
      buffer + counter after T cycles;
    * what a receiver HOLDS (`_costs`, maxsum.py:294, 466: stored at the top of on_new_cycle, so
      the messages of cycle T - 1, start messages included) == the buffer after T - 1 cycles; an
      edge nothing travelled on yet is all zeros there ("zero == not received")."""
    v2f, f2v, cv, cf = at_T
    np.testing.assert_array_equal(cf, ref["count_f2v"])
    np.testing.assert_array_equal(cv, ref["count_v2f"])
    def there(key):  # (fixtures written before the masks existed hold no NaN messages)
        return ref["has_" + key] if "has_" + key in ref else ~np.isnan(ref[key])
    for buf, key in ((f2v, "sent_f2v"), (v2f, "sent_v2f")):
        sent = there(key)
        assert sent.all()          # after one cycle every edge has sent (approx_match(None) is False)
        np.testing.assert_array_equal(buf[sent], ref[key][sent], err_msg=key)
    pv2f, pf2v, _, _ = before_T
    for buf, key in ((pv2f, "held_v2f"), (pf2v, "held_f2v")):
        held = there(key)
        np.testing.assert_array_equal(buf[held], ref[key][held], err_msg=key)
        assert not buf[~held].any(), key


def check_golden(graph, params_kw, meta, ref_idx, ref_cost, lib_path=None, **extra):
    """Engine against the reference's own result stored in a golden fixture:
    final assignment identical, costs within 1e-5 (the north-star tolerance); in f64 every
    message the reference's computations sent / hold and every send counter bit for bit."""
    p = Params(**{**params_kw, **extra})
    eng = MaxSumEngine(graph, p, lib_path=lib_path)
    ref_msgs = meta.get("ref_messages") if p.dtype == "f64" and meta["T"] >= 1 else None
    if ref_msgs is not None:
        eng.run(meta["T"] - 1)
        before = eng.messages()
        eng.run(1)
        assert_messages_equal_reference(ref_msgs, eng.messages(), before)
    else:
        eng.run(meta["T"])
    idx, belief = eng.assignment()
    np.testing.assert_array_equal(idx, ref_idx)
    ok = ~np.isnan(ref_cost)
    np.testing.assert_allclose(belief[ok], ref_cost[ok], rtol=1e-5, atol=1e-5)
    cost, viol = eng.eval_cost()
    assert viol == meta["violation"]
    assert abs(cost - meta["cost"]) <= 1e-5 * max(1.0, abs(meta["cost"]))
    eng.close()


# (name, graph factory, Params kwargs) -- small seeded instances covering every
# kernel class: register binary/unary D in {2,3,4}, lane-grid / workgroup / generic factors (mixed
# domains, arity 1..3, D=5..8), register variables deg<=4 / <=8, generic
# variables (deg>8, D>4), isolated variables, initial values.
def parity_cases():
    def with_init(g, seed):
        rng = np.random.default_rng(seed)
        init = np.where(rng.random(g.n_vars) < 0.3, rng.integers(0, 2, g.n_vars), -1)
        g.init_idx = init.astype(np.int32)
        return g

    def with_inf(g, seed):
        rng = np.random.default_rng(seed)
        t = g.tables.copy()
        t[rng.random(t.shape[0]) < 0.15] = np.inf
        g.tables = t
        return g

    def hard(g, seed, frac, value, what="tables"):
        """Hard constraints as the reference writes them (dcop.py:352-365: +-inf entries; YAML `inf`):
        a fraction of the table entries -- or of the variables' own costs, so that integer tables
        stay in their narrow storage type -- becomes +-inf, and costs_for_factor's mean
        (maxsum.py:671-674) turns inf - inf into NaN."""
        rng = np.random.default_rng(seed)
        a = getattr(g, what).copy()
        a[rng.random(a.shape[0]) < frac] = value
        setattr(g, what, a)
        return g

    def hub(seed, nf=40, n=60, colors=3):
        # one variable in nf more factors (packed class up to degree 64, hub class above -- round 5: wide class up to 256,
        # a thread per variable beyond), some isolated variables
        g = G.random_coloring(n, avg_degree=3, n_colors=colors, seed=seed)
        rng = np.random.default_rng(seed)
        others = rng.choice(np.arange(1, n - 10), size=nf, replace=False)
        edge_var = np.concatenate([g.edge_var, np.stack([np.zeros(nf, int), others], 1).reshape(-1)])
        rowptr = np.concatenate([g.factor_rowptr, g.factor_rowptr[-1] + 2 * np.arange(1, nf + 1)])
        tables = np.concatenate([g.tables, rng.integers(0, 10, nf * colors * colors).astype(float)])
        toff = np.concatenate([g.table_off, g.table_off[-1] + colors * colors * np.arange(1, nf + 1)])
        from pydcop_amd.graph import FlatGraph
        dom = np.concatenate([g.dom_size, [3, 3, 2]])  # 3 isolated variables
        cost = np.concatenate([g.var_cost, rng.uniform(0, 1, 8)])
        vr, ve = FlatGraph.var_side_from_edges(edge_var, dom.shape[0])
        return FlatGraph(dom_size=dom, var_cost=cost, factor_rowptr=rowptr, edge_var=edge_var,
                         table_off=toff, tables=tables, var_rowptr=vr, var_edges=ve).validate()

    def scaled(g, factor, neg_zero=False):
        t = g.tables * factor
        if neg_zero:
            t[t == 0] = -0.0     # an entry no integer type holds (the sign of the zero)
        g.tables = t
        return g

    return [
        ("coloring3_soft", lambda: G.random_coloring(300, seed=1), {}),
        # table storage types (layout.h TabType): quarters fit f32 but no integer type; max mode
        # negates on load; a -0.0 entry; hard 1000 * I is i16
        ("tables_quarters_max", lambda: scaled(G.random_coloring(200, seed=31), 0.25), {"mode": "max"}),
        ("tables_neg_zero", lambda: scaled(G.random_coloring(150, n_colors=2, seed=32), 1.0, True),
         {"mode": "max", "start_messages": "all"}),
        ("tables_i8_max_unary", lambda: G.ising_grid(8, 6, seed=33, bin_range=1.6, un_range=0.05), {"mode": "max"}),
        ("coloring3_hard_all", lambda: G.random_coloring(300, seed=2, variant="hard"),
         {"start_messages": "all", "damping_nodes": "vars"}),
        ("coloring2_deg6_max", lambda: G.random_coloring(200, avg_degree=6, n_colors=2, seed=3),
         {"mode": "max", "start_messages": "leafs_vars", "damping_nodes": "factors"}),
        ("coloring4_none", lambda: G.random_coloring(200, n_colors=4, seed=4),
         {"damping_nodes": "none", "stability": 0.01}),
        ("coloring5_generic", lambda: G.random_coloring(120, n_colors=5, seed=5), {"damping": 0.8}),
        ("ising", lambda: G.ising_grid(9, 7, seed=6), {}),
        ("mixed", lambda: G.random_mixed(60, 90, seed=7), {}),
        ("mixed_max_all", lambda: G.random_mixed(60, 90, seed=8), {"mode": "max", "start_messages": "all"}),
        ("meeting", lambda: G.meeting_like(20, dom=6, seed=9), {"mode": "max"}),
        ("hub_isolated", lambda: hub(10), {}),
        ("coloring3_deg14", lambda: G.random_coloring(120, avg_degree=14, seed=12), {}),
        ("coloring2_deg30_max", lambda: G.random_coloring(90, avg_degree=30, n_colors=2, seed=13),
         {"mode": "max", "start_messages": "leafs_vars"}),
        ("coloring4_deg20_all", lambda: G.random_coloring(80, avg_degree=20, n_colors=4, seed=14),
         {"start_messages": "all", "damping_nodes": "vars"}),
        ("init_values", lambda: with_init(G.random_coloring(100, n_colors=2, seed=11), 11), {}),
        # workgroup-per-factor kernel (arity 2..4, 64 <= R <= 1024)
        ("nary_meeting_d8", lambda: G.meeting_like(30, n_factors=25, dom=8, arity=3, seed=15), {"mode": "max"}),
        ("nary_meeting_d24", lambda: G.meeting_like(12, n_factors=6, dom=24, arity=3, seed=16),
         {"mode": "max", "start_messages": "all"}),
        ("nary_binary_d70", lambda: G.meeting_like(16, n_factors=12, dom=70, arity=2, seed=17), {}),
        ("nary_arity4_d5", lambda: G.meeting_like(25, n_factors=15, dom=5, arity=4, seed=18),
         {"damping_nodes": "factors"}),
        # wave-per-variable kernel, large LDS footprint: deg * D > 128 and degree > 64
        ("wide_coloring6_deg30", lambda: G.random_coloring(60, avg_degree=30, n_colors=6, seed=20), {}),
        ("wide_hub_deg100", lambda: hub(21, nf=100, n=150), {"start_messages": "leafs_vars"}),
        # corners: singleton domains, arity 6 (generic factor), a 300-value domain (generic
        # variable), tables with +inf entries (hard constraints as the reference writes them)
        ("corner_singletons_arity6", lambda: G.random_mixed(30, 25, seed=22, max_arity=6,
                                                            dom_choices=(1, 2, 3)), {}),
        ("corner_domain300", lambda: G.random_mixed(6, 8, seed=23, max_arity=2, dom_choices=(300, 3)),
         {"mode": "max"}),
        ("corner_inf_tables", lambda: with_inf(G.random_coloring(150, seed=24), 24), {}),
        ("nary_mixed_dims", lambda: G.random_mixed(40, 50, seed=19, max_arity=4, dom_choices=(3, 7, 10, 12)), {}),
        # hard constraints (+-inf, then NaN messages) through the LDS kernels: the workgroup-per-factor
        # reductions (inline v_min_f64, DPP / permlane moves, LDS minima on integer keys) and the wide
        # variable kernel's chains -- the register classes have corner_inf_tables above
        ("hard_nary_d8_max", lambda: hard(G.meeting_like(30, n_factors=25, dom=8, arity=3, seed=41), 41, 0.97, -np.inf),
         {"mode": "max"}),
        ("hard_nary_d8_min_all", lambda: hard(G.meeting_like(30, n_factors=25, dom=8, arity=3, seed=42), 42, 0.97, np.inf),
         {"start_messages": "all"}),
        ("hard_nary_d24_max", lambda: hard(G.meeting_like(12, n_factors=6, dom=24, arity=3, seed=43), 43, 0.995, -np.inf),
         {"mode": "max"}),
        ("hard_nary_d24_i8_varcost", lambda: hard(G.meeting_like(12, n_factors=6, dom=24, arity=3, seed=44), 44, 0.4,
                                                  -np.inf, "var_cost"), {"mode": "max", "start_messages": "all"}),
        ("hard_nary_d8_i8_varcost_min", lambda: hard(G.meeting_like(30, n_factors=25, dom=8, arity=3, seed=45), 45, 0.3,
                                                     np.inf, "var_cost"), {"damping_nodes": "factors"}),
        ("hard_wide_coloring6_deg30", lambda: hard(G.random_coloring(60, avg_degree=30, n_colors=6, seed=46), 46, 0.6, np.inf),
         {}),
        # lane-grid kernel of the binary / unary factors beyond the register classes (bin_box.h): every storage type,
        # every lane grid (4 / 16 / 64 lanes), unequal domains, unary tables, the reference's own PEAV model
        ("bin2_coloring8_i8", lambda: G.random_coloring(150, n_colors=8, seed=51), {}),
        ("bin2_coloring8_hard_i16_max_all", lambda: G.random_coloring(120, n_colors=8, seed=52, variant="hard"),
         {"mode": "max", "start_messages": "all"}),
        ("bin2_coloring7_quarters_f32", lambda: scaled(G.random_coloring(120, n_colors=7, seed=53), 0.25),
         {"damping_nodes": "factors"}),
        ("bin2_coloring6_float_max", lambda: scaled(G.random_coloring(120, n_colors=6, seed=54), 0.37), {"mode": "max"}),
        ("bin2_peav_slots10", lambda: G.peav_like(40, 25, slots=10, max_length=4, max_resources_event=4, seed=55), {"mode": "max"}),
        ("bin2_peav_slots23", lambda: G.peav_like(24, 14, slots=23, max_length=7, max_resources_event=4, seed=56),
         {"mode": "max", "start_messages": "leafs_vars"}),
        ("bin2_domains_to_64", lambda: G.random_mixed(16, 22, seed=57, max_arity=2, dom_choices=(33, 40, 48, 64, 5, 13, 21, 27)),
         {"start_messages": "all"}),
        ("bin2_domains_to_64_int_max", lambda: G.random_mixed(16, 22, seed=58, max_arity=2, float_tables=False,
                                                              dom_choices=(1, 2, 9, 17, 29, 37, 50, 64)), {"mode": "max"}),
        ("hard_bin2_coloring8_inf", lambda: hard(G.random_coloring(100, n_colors=8, seed=59), 59, 0.5, np.inf), {}),
        ("hard_bin2_coloring6_neg_inf_max", lambda: hard(G.random_coloring(100, n_colors=6, seed=60), 60, 0.5, -np.inf),
         {"mode": "max", "start_messages": "all"}),
        # one-wave-per-factor box kernel on tables its lane grid overhangs (round 5: dimensions that are no multiples of the box)
        ("box_overhang_24_23_22", lambda: G.meeting_hetero(14, n_factors=8, doms=(24, 23, 22), seed=61), {"mode": "max"}),
        ("box_overhang_24_20", lambda: G.meeting_hetero(14, n_factors=8, doms=(24, 20, 21), seed=62),
         {"mode": "max", "start_messages": "all"}),
        ("box_overhang_small", lambda: G.meeting_hetero(30, n_factors=24, doms=(10, 11, 12, 8, 7), seed=63), {"mode": "max"}),
        # int16 tables on the 6 x 6 x 6 box shape: a lane's record (108 dwords) in two passes (round 5)
        ("box_i16_d24_two_passes", lambda: G.meeting_like(12, n_factors=6, dom=24, arity=3, seed=67, penalty=1000.0), {"mode": "max"}),
        ("box_i16_overhang_min_all", lambda: G.meeting_hetero(14, n_factors=8, doms=(24, 23, 22), seed=68, penalty=3000.0),
         {"start_messages": "all"}),
        ("hard_box_overhang_varcost", lambda: hard(G.meeting_hetero(14, n_factors=8, doms=(24, 23, 22), seed=64), 64, 0.4,
                                                   -np.inf, "var_cost"), {"mode": "max", "start_messages": "all"}),
        # arity 3 / 4 with FEWER than 64 entries per value of the first variable: one wave with idle lanes (round 5; generic before)
        ("nary_small_rows_5x5x5", lambda: G.meeting_like(30, n_factors=20, dom=5, arity=3, seed=65), {"mode": "max"}),
        ("nary_small_rows_mixed", lambda: G.random_mixed(40, 40, seed=66, max_arity=4, dom_choices=(3, 4, 5, 7)), {"start_messages": "all"}),
        # hub class (round 6, kernels.h variable_hub): a wave per 64 outgoing edges of one variable, a lane per edge.  Degrees the
        # scale-free generator of the reference reaches (graphcoloring.py:322-340: ~1 000 at 100k variables, ~3 400 at 1M), one
        # tile and several per value of d (HUB_TILE = 512), domains of the packed-on-8 class, a wide domain at a degree the wide
        # class cannot stage (deg * D > 1024), hard constraints, initial values
        ("hub_deg300", lambda: hub(71, nf=300, n=400), {}),
        ("hub_deg1100_max_all", lambda: hub(72, nf=1100, n=1300), {"mode": "max", "start_messages": "all"}),
        ("hub_deg3400", lambda: hub(73, nf=3400, n=3600), {"start_messages": "leafs_vars", "damping_nodes": "vars"}),
        ("hub_deg513_d2", lambda: hub(74, nf=510, n=640, colors=2), {"damping_nodes": "factors"}),
        ("hub_deg200_d6", lambda: hub(75, nf=200, n=300, colors=6), {"mode": "max"}),
        ("hub_deg90_d24", lambda: hub(76, nf=90, n=140, colors=24), {"mode": "max", "start_messages": "all"}),
        ("hub_scalefree_2000", lambda: G.scalefree_coloring(2000, m=2, seed=77), {}),
        ("hub_scalefree_m3_d4_init", lambda: with_init(G.scalefree_coloring(1500, m=3, n_colors=4, seed=78), 78), {"damping": 0.3}),
        ("hard_hub_deg300_inf", lambda: hard(hub(79, nf=300, n=400), 79, 0.3, np.inf), {}),
        ("hard_hub_deg150_varcost_neg_inf_max", lambda: hard(hub(80, nf=150, n=220), 80, 0.2, -np.inf, "var_cost"),
         {"mode": "max", "start_messages": "all"}),
        # round 6: the reference's `generate secp` (generators.secp_like == secp.py's expressions): D = 5, unary real tables, model
        # constraints of arity 3..4 in {0, 10000}, rules of arity 1..3; with --max_model_size 4 ARITY 5 -- the workgroup-per-
        # factor kernels at A = 5 (before: thread per edge)
        ("secp_small", lambda: G.secp_like(60, 40, 50, seed=81), {}),
        ("secp_small_m4_all", lambda: G.secp_like(50, 40, 40, max_model_size=4, seed=82, unary_noise=0.01), {"start_messages": "all"}),
        ("nary_arity5_d5_max", lambda: G.meeting_like(30, n_factors=12, dom=5, arity=5, seed=83), {"mode": "max"}),
        ("nary_arity5_mixed_float", lambda: G.random_mixed(30, 14, seed=84, max_arity=5, dom_choices=(3, 4, 5, 6)), {"damping_nodes": "factors"}),
        ("hard_secp_small_m4_inf", lambda: hard(G.secp_like(50, 40, 40, max_model_size=4, seed=85), 85, 0.2, np.inf), {}),
        ("hard_wide_coloring6_deg30_max_all", lambda: hard(G.random_coloring(60, avg_degree=30, n_colors=6, seed=47), 47, 0.6,
                                                           -np.inf), {"mode": "max", "start_messages": "all"}),
        # round 6: the workgroup-per-factor kernel in PASSES (kernels.h k_factor_nary<.., MULTI>): more than 1 024 entries per value
        # of the first variable -- arity 3 over 33+ values, arity 4 over 11+, arity 5 over 6+ -- and arity 6, a SECP instance
        # generated with --max_model_size 5 among them (thread per edge before)
        ("multi_arity3_d40_max", lambda: G.meeting_like(10, n_factors=5, dom=40, arity=3, seed=86), {"mode": "max"}),
        ("multi_arity3_d64_float_all", lambda: G.meeting_like(8, n_factors=3, dom=64, arity=3, seed=87, float_tables=True),
         {"mode": "max", "start_messages": "all"}),
        ("multi_arity4_d11", lambda: G.meeting_like(12, n_factors=4, dom=11, arity=4, seed=88), {"damping_nodes": "factors"}),
        ("multi_arity5_d6_max", lambda: G.meeting_like(14, n_factors=4, dom=6, arity=5, seed=89), {"mode": "max"}),
        ("multi_arity6_secp_m5", lambda: G.secp_like(50, 40, 40, max_model_size=5, seed=90), {}),
        ("multi_arity6_mixed_dims", lambda: G.random_mixed(30, 12, seed=91, max_arity=6, dom_choices=(2, 3, 4, 7)), {"start_messages": "all"}),
        ("hard_multi_arity3_d40_neg_inf", lambda: hard(G.meeting_like(10, n_factors=5, dom=40, arity=3, seed=92), 92, 0.9, -np.inf),
         {"mode": "max"}),
    ]


def check_table_updates(oracle_mod, graph, params: Params, lib_path=None, seed=0):
    """mxs_update_factor_table (maxsum_dynamic.py:80-104, change_factor_function): new
    tables for a tenth of the factors in the middle of a run, engine == oracle before and
    after, and the solution cost follows the new tables."""
    rng = np.random.default_rng(seed)
    eng = MaxSumEngine(graph, params, lib_path=lib_path)
    ora = oracle_mod.OracleMaxSum(graph, params)
    eng.run(4), ora.run(4)
    picks = rng.choice(graph.n_factors, size=max(1, graph.n_factors // 10), replace=False)
    for f in picks:
        n = int(graph.table_off[f + 1] - graph.table_off[f])
        t = rng.integers(-5, 15, n).astype(np.float64)
        eng.update_factor_table(int(f), t)
        ora.update_factor_table(int(f), t)
    for n in (1, 7):
        eng.run(n), ora.run(n)
        for x, y in zip(eng.messages(), ora.messages()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(eng.assignment()[0], ora.assignment()[0])
        np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1])
        ce, co = eng.eval_cost(), ora.eval_cost()
        assert ce[1] == co[1] and abs(ce[0] - co[0]) <= 1e-9 * max(1.0, abs(co[0]))
    # values that no narrow table type holds (a class storing i8 / i16 / f32 records falls back
    # to its full-width image, layout.h TabType), then small integers again
    for k, f in enumerate(picks[:3]):
        n = int(graph.table_off[f + 1] - graph.table_off[f])
        t = rng.integers(-5, 15, n).astype(np.float64)
        t[0] = (300.0, 1e6 + 0.25, 0.1)[k]
        eng.update_factor_table(int(f), t)
        ora.update_factor_table(int(f), t)
        eng.run(2), ora.run(2)
        for x, y in zip(eng.messages(), ora.messages()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1])
        ce, co = eng.eval_cost(), ora.eval_cost()
        assert ce[1] == co[1] and abs(ce[0] - co[0]) <= 1e-9 * max(1.0, abs(co[0]))
    import pytest
    from pydcop_amd.engine import MaxSumGpuError
    with pytest.raises(MaxSumGpuError):
        eng.update_factor_table(0, np.zeros(int(graph.table_off[1] - graph.table_off[0]) + 1))
    with pytest.raises(MaxSumGpuError):
        eng.update_factor_table(graph.n_factors, np.zeros(1))
    eng.close()
    ora.close()
