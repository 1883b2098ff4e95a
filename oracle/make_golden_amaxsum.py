"""Generate tests/golden/amaxsum/*.npz from the REFERENCE itself (build container only): what the
reference's own amaxsum computations (pydcop/algorithms/amaxsum.py) hold after G generations of
FIFO delivery (oracle/ref_harness.run_reference_amaxsum; G = -1: until no message is left) --
selected value index + cost per variable, the messages of every generation's size, and
DCOP.solution_cost of the assignment.  TEST INFRASTRUCTURE.

    python -m oracle.make_golden_amaxsum
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_harness import REFERENCE_ROOT, flat_to_dcop, install_shims, run_reference_amaxsum  # noqa: E402
from pydcop_amd import generators as G  # noqa: E402
from pydcop_amd.compile import compile_computation_graph  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "amaxsum")
GRAPH_FIELDS = ("dom_size", "var_cost", "factor_rowptr", "edge_var", "table_off", "tables", "var_rowptr", "var_edges")


def save(name, graph, mode, params, gens, vals, costs, info, sol):
    idx = np.array([graph.domains[i].index(vals[n]) for i, n in enumerate(graph.var_names)], dtype=np.int32)
    cost = np.array([0.0 if costs[n] is None else costs[n] for n in graph.var_names])
    meta = dict(name=name, mode=mode, params=params, generations=gens, delivered=info["delivered"],
                pending=info["pending"], generation_sizes=info["generation_sizes"], violation=int(sol[0]),
                cost=float(sol[1]), var_names=graph.var_names, domains=[[str(x) for x in d] for d in graph.domains])
    arrays = {k: getattr(graph, k) for k in GRAPH_FIELDS}
    if graph.init_idx is not None:
        arrays["init_idx"] = graph.init_idx
    np.savez_compressed(os.path.join(OUT, name + ".npz"), ref_idx=idx, ref_cost=cost,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"{name}: V={graph.n_vars} F={graph.n_factors} generations={gens} delivered={info['delivered']} "
          f"cost={sol[1]:.6g} viol={sol[0]}")


def main():
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    from pydcop.computations_graph import factor_graph
    from pydcop.dcop.yamldcop import load_dcop_from_file
    inst = os.path.join(REFERENCE_ROOT, "tests", "instances")
    for fname, params in (("graph_coloring1.yaml", {"start_messages": "leafs_vars"}),
                          ("secp_simple1.yaml", {"start_messages": "all"}),
                          ("graph_coloring_tuto.yaml", {"start_messages": "leafs_vars"}),
                          ("graph_coloring_3agts_10vars.yaml", {"start_messages": "all", "damping_nodes": "vars"})):
        dcop = load_dcop_from_file([os.path.join(inst, fname)])
        cg = factor_graph.build_computation_graph(dcop)
        graph = compile_computation_graph(cg)
        for gens in (6, -1):
            vals, costs, info = run_reference_amaxsum(dcop, gens, params, cg=cg, max_messages=2_000_000)
            sol = dcop.solution_cost(vals, float("inf"))
            save(f"yaml_{fname.split('.')[0]}_G{gens if gens >= 0 else 'end'}", graph, dcop.objective, params, gens,
                 vals, costs, info, sol)
    for name, g, mode, params in (
            ("coloring60", G.random_coloring(60, seed=41), "min", {"start_messages": "leafs_vars"}),
            ("coloring_hard40", G.random_coloring(40, seed=42, variant="hard"), "min", {"start_messages": "all"}),
            ("mixed_max", G.random_mixed(20, 28, seed=43), "max", {"start_messages": "leafs_vars", "damping_nodes": "factors"}),
            ("ising_5x5", G.ising_grid(5, 5, seed=44), "min", {"start_messages": "all", "damping": 0.7})):
        dcop, cg = flat_to_dcop(g, mode)
        for gens in (8, -1):
            if gens < 0 and name == "coloring_hard40":
                continue  # hard 1000 * I tables keep it talking: no quiescence to record
            vals, costs, info = run_reference_amaxsum(dcop, gens, params, cg=cg, max_messages=2_000_000)
            sol = dcop.solution_cost(vals, float("inf"))
            save(f"syn_{name}_G{gens if gens >= 0 else 'end'}", g, mode, params, gens, vals, costs, info, sol)


if __name__ == "__main__":
    main()
