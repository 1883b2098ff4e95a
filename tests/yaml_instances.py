"""Test helper: write a FlatGraph as a pyDCOP YAML DCOP (pydcop/dcop/yamldcop.py:96-313: domains,
variables with `cost_function`, `extensional` constraints, agents) + a distribution file that
puts every computation of the factor graph on ONE agent (pydcop/distribution/yamlformat.py:44-59)
-- the form in which a generated instance reaches `pydcop solve --algo maxsum_gpu -d dist.yaml`."""
import numpy as np


def write_coloring_yaml(graph, dcop_path, dist_path=None, objective="min", agent="a0"):
    """`graph`: a FlatGraph with names (generators.random_coloring(..., names=True)); integer
    domains 0..D-1; tables with few distinct values (the YAML groups assignments by cost).
    The numbers are written with repr(), so the loaded DCOP holds the same f64 bits."""
    g = graph
    out = ["name: generated", f"objective: {objective}", "", "domains:"]
    sizes = sorted({int(d) for d in g.dom_size})
    for d in sizes:
        out.append(f"  d{d}:\n    values: [{', '.join(str(x) for x in range(d))}]")
    out.append("\nvariables:")
    off = g.cost_off
    for i, name in enumerate(g.var_names):
        D = int(g.dom_size[i])
        costs = ", ".join(repr(float(c)) for c in g.var_cost[off[i]:off[i] + D])
        out.append(f"  {name}:\n    domain: d{D}\n    cost_function: '[{costs}][{name}]'")
    out.append("\nconstraints:")
    for f, name in enumerate(g.factor_names):
        e0, e1 = int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])
        scope = [int(v) for v in g.edge_var[e0:e1]]
        shape = tuple(int(g.dom_size[v]) for v in scope)
        table = np.asarray(g.tables[g.table_off[f]:g.table_off[f + 1]]).reshape(shape)
        groups = {}
        for idx in np.ndindex(*shape):
            groups.setdefault(float(table[idx]), []).append(" ".join(str(x) for x in idx))
        out.append(f"  {name}:\n    type: extensional\n    variables: [{', '.join(g.var_names[v] for v in scope)}]\n    values:")
        for val, asses in groups.items():
            key = repr(int(val)) if val == int(val) else repr(val)
            out.append(f"      {key}: '{' | '.join(asses)}'")
    out.append(f"\nagents:\n  {agent}:\n    capacity: 1000000000\n")
    with open(dcop_path, "w") as fh:
        fh.write("\n".join(out))
    if dist_path:
        names = list(g.var_names) + list(g.factor_names)
        with open(dist_path, "w") as fh:
            fh.write(f"distribution:\n  {agent}: [{', '.join(names)}]\n")
    return dcop_path
