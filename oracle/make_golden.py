"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build
container only; needs /root/reference).  TEST INFRASTRUCTURE.

Each fixture holds a compiled FlatGraph, the algorithm parameters, a cycle
count T and what the reference's own MaxSum computations
(pydcop/algorithms/maxsum.py) hold after exactly T cycles -- selected value
index + cost per variable, every sender's `_prev_messages` (last sent message
and count) and every receiver's `_costs` (ref_harness.reference_message_state)
-- plus DCOP.solution_cost of that assignment (pydcop/dcop/dcop.py:308-367).

    python -m oracle.make_golden
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import (REFERENCE_ROOT, flat_to_dcop, install_shims,  # noqa: E402
                                reference_message_state, run_reference_maxsum)
from pydcop_amd import generators as G  # noqa: E402
from pydcop_amd.compile import compile_computation_graph  # noqa: E402

OUT = os.environ.get("GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
GRAPH_FIELDS = ("dom_size", "var_cost", "factor_rowptr", "edge_var", "table_off",
                "tables", "var_rowptr", "var_edges")


def run(dcop, T, params, cg):
    """-> (values, costs, message state of the reference's computations after T cycles)"""
    vals, costs, comps = run_reference_maxsum(dcop, T, params, cg=cg, return_comps=True)
    return vals, costs, comps


def save(name, graph, mode, params, T, vals, costs, sol, comps=None):
    idx = np.array([graph.domains[i].index(vals[n]) for i, n in enumerate(graph.var_names)],
                   dtype=np.int32)
    cost = np.array([np.nan if costs[n] is None else costs[n] for n in graph.var_names])
    meta = dict(name=name, mode=mode, params=params, T=T, violation=int(sol[0]),
                cost=float(sol[1]), var_names=graph.var_names,
                domains=[[str(x) for x in d] for d in graph.domains],
                values=[str(vals[n]) for n in graph.var_names])
    arrays = {k: getattr(graph, k) for k in GRAPH_FIELDS}
    if graph.init_idx is not None:
        arrays["init_idx"] = graph.init_idx
    if comps is not None:
        for k, a in reference_message_state(comps, graph).items():
            arrays["ref_" + k] = a
    np.savez_compressed(os.path.join(OUT, name + ".npz"), ref_idx=idx, ref_cost=cost,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"{name}: V={graph.n_vars} F={graph.n_factors} E={graph.n_edges} T={T} "
          f"cost={sol[1]:.6g} viol={sol[0]}")


def yaml_cases():
    install_shims()
    from pydcop.computations_graph import factor_graph
    from pydcop.dcop.yamldcop import load_dcop_from_file
    inst = os.path.join(REFERENCE_ROOT, "tests", "instances")
    cases = [("graph_coloring1.yaml", (5, 20, 50)), ("secp_simple1.yaml", (5, 20)),
             ("graph_coloring_tuto.yaml", (5, 20)), ("graph_coloring_tuto_max.yaml", (20,)),
             ("graph_coloring_10_4_15_0.1.yml", (20,)), ("graph_coloring_eq.yaml", (20,)),
             ("graph_coloring_3agts_10vars.yaml", (20,)), ("graph_coloring1_func.yaml", (20,))]
    for fname, Ts in cases:
        dcop = load_dcop_from_file([os.path.join(inst, fname)])
        cg = factor_graph.build_computation_graph(dcop)
        graph = compile_computation_graph(cg)
        for T in Ts:
            vals, costs, comps = run(dcop, T, None, cg)
            sol = dcop.solution_cost(vals, float("inf"))
            save(f"yaml_{fname.split('.')[0].replace('.', '_')}_T{T}", graph, dcop.objective,
                 {}, T, vals, costs, sol, comps)


def synthetic_cases():
    cases = [
        ("coloring_soft_200", G.random_coloring(200, seed=0), "min", {}, (1, 3, 25, 60)),
        ("coloring_hard_200", G.random_coloring(200, seed=1, variant="hard"), "min", {}, (25,)),
        ("coloring_soft_200_all_vars", G.random_coloring(200, seed=2), "min",
         {"start_messages": "all", "damping_nodes": "vars", "damping": 0.7}, (7, 25)),
        ("coloring_soft_200_lv_none_max", G.random_coloring(200, seed=3), "max",
         {"start_messages": "leafs_vars", "damping_nodes": "none", "stability": 0.01}, (7, 25)),
        ("coloring_d4_deg6_150", G.random_coloring(150, avg_degree=6, n_colors=4, seed=4), "min",
         {"damping_nodes": "factors"}, (25,)),
        ("ising_8x8", G.ising_grid(8, 8, seed=0), "min", {}, (3, 25)),
        ("mixed_40_60", G.random_mixed(40, 60, seed=0), "min", {}, (3, 25)),
        ("mixed_40_60_max_all", G.random_mixed(40, 60, seed=1), "max",
         {"start_messages": "all"}, (25,)),
        ("meeting_16_d5", G.meeting_like(16, dom=5, seed=0), "max", {}, (3, 25)),
        ("meeting_10_d8_a3", G.meeting_like(10, n_factors=8, dom=8, seed=1), "max",
         {"damping_nodes": "vars"}, (10,)),
    ]
    for name, graph, mode, params, Ts in cases:
        dcop, cg = flat_to_dcop(graph, mode)
        for T in Ts:
            vals, costs, comps = run(dcop, T, params, cg)
            sol = dcop.solution_cost(vals, float("inf"))
            save(f"syn_{name}_T{T}", graph, mode, params, T, vals, costs, sol, comps)


def generator_cases():
    """Instances produced by the reference's OWN generators
    (pydcop/commands/generators/{ising,graphcoloring,meetingscheduling}.py), i.e. the DCOP
    objects `pydcop generate ...` would write, solved by the reference's Max-Sum."""
    import random
    install_shims()
    from pydcop.commands.generators import graphcoloring as gc
    from pydcop.commands.generators import ising, meetingscheduling as ms
    from pydcop.computations_graph import factor_graph
    from pydcop.dcop.dcop import DCOP
    from pydcop.dcop.objects import Variable, VariableDomain

    def emit(name, dcop, Ts, params=None):
        cg = factor_graph.build_computation_graph(dcop)
        graph = compile_computation_graph(cg)
        for T in Ts:
            vals, costs, comps = run(dcop, T, params or {}, cg)
            sol = dcop.solution_cost(vals, float("inf"))
            save(f"gen_{name}_T{T}", graph, dcop.objective, params or {}, T, vals, costs, sol, comps)

    random.seed(20260923)
    # pydcop generate ising --row_count 5 --col_count 4 [--intentional]   (ising.py:274-331)
    for label, extensive in (("ext", True), ("int", False)):
        dcop, _, _ = ising.generate_ising(5, 4, 1.6, 0.05, extensive, True, False, False)
        emit(f"ising_5x4_{label}", dcop, (3, 20))
    # pydcop generate graph_coloring -v 24 -c 3 --graph random --p_edge 0.15 --soft [--intentional]
    # and --graph scalefree --m_edge 2  (graphcoloring.py:238-306, 355-413)
    def coloring(graph, soft, intentional, colors=3):
        domain = VariableDomain("colors", "color", gc.COLORS[:colors])
        variables = {node: Variable(f"v{i:02d}", domain) for i, node in enumerate(sorted(graph.nodes))}
        make = gc.generate_soft_constraints if soft else gc.generate_hard_constraints
        return DCOP("gc", domains={"colors": domain},
                    variables={v.name: v for v in variables.values()}, agents={},
                    constraints=make(graph, variables, intentional))
    emit("coloring_random_soft_ext", coloring(gc.generate_random_graph(24, 0.15, True), True, False), (3, 20))
    emit("coloring_random_hard_int", coloring(gc.generate_random_graph(24, 0.15, True), False, True), (20,),
         {"damping_nodes": "vars"})
    emit("coloring_scalefree_hard", coloring(gc.generate_scalefree_graph(30, 2, True), False, True, 4), (20,),
         {"start_messages": "leafs_vars"})
    # pydcop generate meetings --slots_count 5 --events_count 5 --resources_count 4 ...
    # (meetingscheduling.py:210-226, 317-365: PEAV model, objective max)
    slots, events, resources = ms.generate_problem_definition(5, 4, 8, 5, 2, 3)
    variables, constraints, _ = ms.peav_model(slots, events, resources, 8 * 5 * 4)
    dcop = DCOP("meetings", objective="max",
                domains={v.domain.name: v.domain for v in variables.values()},
                variables={v.name: v for v in variables.values()}, constraints=constraints, agents={})
    emit("meetings_peav", dcop, (5, 20))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--only-generators" not in sys.argv:
        yaml_cases()
        synthetic_cases()
    generator_cases()
