"""O(E) synthetic instance generators producing FlatGraphs directly.

The reference's generators (pydcop/commands/generators/*.py) build networkx
graphs and DCOP objects and do not scale (nx.gnp_random_graph is O(N^2),
build_computation_graph O(V*F)); these follow their *instance conventions* only:

  random_coloring  soft: extensional tables of `randint(0, 9)` costs
                   (graphcoloring.py:355-375); hard: cost 1000 on equal colours
                   (graphcoloring.py:378-413)
  scalefree_coloring  the same tables on a Barabasi-Albert graph (`--graph scalefree`,
                   graphcoloring.py:322-340): hub variables of degree ~ sqrt(n)
  ising_grid       periodic grid, binary [[k,-k],[-k,k]] with k~U(-r, r) and
                   unary [u,-u] with u~U(-ur, ur) kept as real unary factors
                   (ising.py:285, 362-383, 412-420)
  meeting_like     arity-3, D=24 tables, objective max (SURVEY.md section 8d cfg 5)
  peav_like        the reference's OWN meeting-scheduling model (PEAV,
                   meetingscheduling.py:317-365, 441-640): one variable per (resource, event),
                   domains `range(0, slots - length + 2)` (:450-454), binary intra-resource
                   utility / conflict tables (:540-585), binary inter-resource equality tables
                   of 0 / -penalty (:588-599), a unary utility table for a resource with a
                   single event (:497-510); objective max

Symmetry is broken by per-variable unary costs U(0, 0.01) stored as variable
costs: the deterministic stand-in for the reference's unseeded noise
(pydcop/dcop/objects.py:566-567).
"""
import numpy as np

from .graph import FlatGraph


def _finish(dom_size, var_cost, factor_rowptr, edge_var, tables, table_off, names=True):
    n_vars = len(dom_size)
    var_rowptr, var_edges = FlatGraph.var_side_from_edges(edge_var, n_vars)
    g = FlatGraph(dom_size=dom_size, var_cost=var_cost, factor_rowptr=factor_rowptr,
                  edge_var=edge_var, table_off=table_off, tables=tables,
                  var_rowptr=var_rowptr, var_edges=var_edges)
    if names:
        width = len(str(max(n_vars - 1, 1)))
        g.var_names = [f"v{i:0{width}d}" for i in range(n_vars)]
        nf = g.n_factors
        widthf = len(str(max(nf - 1, 1)))
        g.factor_names = [f"c{i:0{widthf}d}" for i in range(nf)]
        g.domains = [list(range(int(d))) for d in g.dom_size]
    return g


def _random_simple_edges(n_vars, n_edges, rng):
    """G(n, m): m distinct unordered pairs without self loops, O(m)."""
    pairs = np.zeros((0, 2), dtype=np.int64)
    seen = np.zeros(0, dtype=np.int64)
    while pairs.shape[0] < n_edges:
        need = n_edges - pairs.shape[0]
        a = rng.integers(0, n_vars, size=int(need * 1.2) + 16)
        b = rng.integers(0, n_vars, size=a.shape[0])
        keep = a != b
        lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
        key = lo * n_vars + hi
        key, first = np.unique(key, return_index=True)
        fresh = ~np.isin(key, seen)
        key = key[fresh]
        order = rng.permutation(key.shape[0])[:need]
        key = key[order]
        seen = np.concatenate([seen, key])
        pairs = np.concatenate([pairs, np.stack([key // n_vars, key % n_vars], axis=1)])
    return pairs[:n_edges]


def _barabasi_albert_edges(n_vars, m, rng):
    """Barabasi-Albert preferential attachment, the construction of nx.barabasi_albert_graph the reference's
    `--graph scalefree` calls (graphcoloring.py:322-340): a star on m + 1 nodes, then every new node attaches to m
    DISTINCT existing nodes drawn with probability proportional to their degree (uniform draws from the list of
    edge endpoints); the node names are shuffled afterwards, as the reference does (:332-339).  O(E)."""
    if not 1 <= m < n_vars:
        raise ValueError("scale-free graph: 1 <= m < n_vars")
    n_edges = m + (n_vars - m - 1) * m
    ends = np.empty(2 * n_edges, dtype=np.int64)       # every edge's two endpoints = the attachment urn
    src = np.empty(n_edges, dtype=np.int64)
    dst = np.empty(n_edges, dtype=np.int64)
    for i in range(m):                                  # the star: node m is the centre
        src[i], dst[i] = i, m
        ends[2 * i], ends[2 * i + 1] = i, m
    ne = m
    u = rng.random(size=(n_vars, 2 * m + 8))            # pre-drawn uniforms; a row per new node
    for node in range(m + 1, n_vars):
        L = 2 * ne
        targets = []
        j = 0
        row = u[node]
        while len(targets) < m:
            if j >= row.shape[0]:
                row, j = rng.random(size=row.shape[0]), 0
            t = int(ends[int(row[j] * L)])
            j += 1
            if t not in targets:
                targets.append(t)
        for t in targets:
            src[ne], dst[ne] = node, t
            ends[2 * ne], ends[2 * ne + 1] = node, t
            ne += 1
    relabel = rng.permutation(n_vars)
    return np.stack([relabel[src], relabel[dst]], axis=1)


def scalefree_coloring(n_vars, m=2, n_colors=3, seed=0, variant="soft", unary_noise=0.01, names=True) -> FlatGraph:
    """Graph colouring on a Barabasi-Albert graph -- what `pydcop generate graph_coloring --graph scalefree
    --m_edge m` emits (graphcoloring.py:322-340): about m * n_vars binary factors, average degree 2 m, and a few
    HUB variables whose degree grows like sqrt(n_vars) (100k variables, m = 2: maximum degree ~ 1 000)."""
    rng = np.random.default_rng(seed)
    return random_coloring(n_vars, n_colors=n_colors, variant=variant, unary_noise=unary_noise, names=names,
                           pairs=_barabasi_albert_edges(n_vars, m, rng), rng=rng)


def random_coloring(n_vars, avg_degree=4, n_colors=3, seed=0, variant="soft",
                    unary_noise=0.01, names=True, pairs=None, rng=None) -> FlatGraph:
    """Random graph colouring: n_vars*avg_degree/2 binary factors on G(n, m) (or on the given `pairs`)."""
    rng = rng if rng is not None else np.random.default_rng(seed)
    if pairs is None:
        n_factors = int(n_vars * avg_degree // 2)
        pairs = _random_simple_edges(n_vars, n_factors, rng)
    n_factors = pairs.shape[0]
    # random orientation of each constraint's scope
    flip = rng.random(n_factors) < 0.5
    pairs = np.where(flip[:, None], pairs[:, ::-1], pairs)
    D = n_colors
    dom_size = np.full(n_vars, D, dtype=np.int32)
    var_cost = rng.uniform(0.0, unary_noise, size=n_vars * D) if unary_noise else np.zeros(n_vars * D)
    factor_rowptr = np.arange(0, 2 * n_factors + 1, 2, dtype=np.int32)
    edge_var = pairs.reshape(-1).astype(np.int32)
    if variant == "soft":
        tables = rng.integers(0, 10, size=(n_factors, D, D)).astype(np.float64)
    elif variant == "hard":
        tables = np.tile(1000.0 * np.eye(D), (n_factors, 1, 1))
    else:
        raise ValueError("variant must be 'soft' or 'hard'")
    table_off = np.arange(0, (n_factors + 1) * D * D, D * D, dtype=np.int64)
    return _finish(dom_size, var_cost, factor_rowptr, edge_var, tables.reshape(-1), table_off, names)


def ising_grid(rows, cols, seed=0, bin_range=1.6, un_range=0.05, names=True) -> FlatGraph:
    """Periodic rows x cols Ising grid, D=2; unary terms are real unary factors."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    r, c = np.divmod(np.arange(n, dtype=np.int64), cols)
    right = r * cols + (c + 1) % cols
    down = ((r + 1) % rows) * cols + c
    me = np.arange(n, dtype=np.int64)
    # per cell: unary factor, horizontal coupling, vertical coupling (keeps locality)
    k_h = rng.uniform(-bin_range, bin_range, size=n)
    k_v = rng.uniform(-bin_range, bin_range, size=n)
    u = rng.uniform(-un_range, un_range, size=n)
    edge_var = np.stack([me, me, right, me, down], axis=1).reshape(-1).astype(np.int32)
    per = np.array([0, 1, 3, 5], dtype=np.int64)
    factor_rowptr = (np.arange(n, dtype=np.int64)[:, None] * 5 + per[None, :3]).reshape(-1)
    factor_rowptr = np.concatenate([factor_rowptr, [5 * n]]).astype(np.int32)
    tab = np.empty((n, 10), dtype=np.float64)
    tab[:, 0], tab[:, 1] = u, -u
    for off, k in ((2, k_h), (6, k_v)):
        tab[:, off + 0], tab[:, off + 1], tab[:, off + 2], tab[:, off + 3] = k, -k, -k, k
    sizes = np.tile(np.array([2, 4, 4], dtype=np.int64), n)
    table_off = np.zeros(3 * n + 1, dtype=np.int64)
    np.cumsum(sizes, out=table_off[1:])
    dom_size = np.full(n, 2, dtype=np.int32)
    var_cost = np.zeros(2 * n)
    return _finish(dom_size, var_cost, factor_rowptr, edge_var, tab.reshape(-1), table_off, names)


def meeting_like(n_vars, n_factors=None, dom=24, arity=3, seed=0, penalty=100.0,
                 unary_noise=0.01, names=True, float_tables=False) -> FlatGraph:
    """Meeting-scheduling-like instance: arity-`arity` factors over D=`dom`
    slot variables; utility `integers(-10, 10)` minus `penalty` when the
    participants do not all pick the same slot.  To be solved with mode 'max'.
    `float_tables`: utilities `uniform(-10, 10)` -- no narrow type holds them, the tables
    are read at full width."""
    rng = np.random.default_rng(seed)
    n_factors = n_factors or n_vars
    scope = np.empty((n_factors, arity), dtype=np.int64)
    for i in range(n_factors):  # distinct variables per factor
        scope[i] = rng.choice(n_vars, size=arity, replace=False) if n_vars < 4096 else 0
    if n_vars >= 4096:
        scope = rng.integers(0, n_vars, size=(n_factors, arity))
        for _ in range(8):  # re-draw the rare duplicates
            s = np.sort(scope, axis=1)
            bad = (np.diff(s, axis=1) == 0).any(axis=1)
            if not bad.any():
                break
            scope[bad] = rng.integers(0, n_vars, size=(int(bad.sum()), arity))
    dom_size = np.full(n_vars, dom, dtype=np.int32)
    var_cost = rng.uniform(0.0, unary_noise, size=n_vars * dom) if unary_noise else np.zeros(n_vars * dom)
    size = dom ** arity
    if float_tables:
        tables = rng.uniform(-10.0, 10.0, size=(n_factors, size))
    else:
        tables = rng.integers(-10, 10, size=(n_factors, size)).astype(np.float64)
    grid = np.indices((dom,) * arity).reshape(arity, -1)
    same = (grid == grid[0]).all(axis=0)
    tables -= np.where(same, 0.0, penalty)[None, :]  # one broadcast pass (x - 0.0 == x exactly)
    factor_rowptr = np.arange(0, arity * n_factors + 1, arity, dtype=np.int32)
    table_off = np.arange(0, (n_factors + 1) * size, size, dtype=np.int64)
    return _finish(dom_size, var_cost, factor_rowptr, scope.reshape(-1).astype(np.int32),
                   tables.reshape(-1), table_off, names)


def meeting_hetero(n_vars, n_factors=None, doms=(24, 23, 22), arity=3, seed=0, penalty=100.0, unary_noise=0.01,
                   names=True) -> FlatGraph:
    """meeting_like with HETEROGENEOUS domains: every variable draws its number of slots from `doms`
    (the PEAV model's `slots - length + 2`, meetingscheduling.py:450-454), integer utilities
    `integers(-10, 10)` minus `penalty` off the all-equal diagonal.  To be solved with mode 'max'."""
    rng = np.random.default_rng(seed)
    n_factors = n_factors or n_vars
    dom_size = rng.choice(np.array(doms), size=n_vars).astype(np.int32)
    scope = np.stack([rng.choice(n_vars, size=arity, replace=False) for _ in range(n_factors)]) if n_vars < 4096 \
        else rng.integers(0, n_vars, size=(n_factors, arity))
    if n_vars >= 4096:
        for _ in range(8):
            bad = (np.diff(np.sort(scope, axis=1), axis=1) == 0).any(axis=1)
            if not bad.any():
                break
            scope[bad] = rng.integers(0, n_vars, size=(int(bad.sum()), arity))
    var_cost = rng.uniform(0.0, unary_noise, size=int(dom_size.sum())) if unary_noise else np.zeros(int(dom_size.sum()))
    sizes = np.prod(dom_size[scope], axis=1).astype(np.int64)
    table_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    tables = rng.integers(-10, 10, size=int(table_off[-1])).astype(np.float64)
    cache = {}
    for f in range(n_factors):
        shape = tuple(int(d) for d in dom_size[scope[f]])
        if shape not in cache:
            grid = np.indices(shape).reshape(arity, -1)
            cache[shape] = np.where((grid == grid[0]).all(axis=0), 0.0, penalty)
        tables[table_off[f]:table_off[f + 1]] -= cache[shape]
    factor_rowptr = np.arange(0, arity * n_factors + 1, arity, dtype=np.int32)
    return _finish(dom_size, var_cost, factor_rowptr, scope.reshape(-1).astype(np.int32), tables, table_off, names)


def random_mixed(n_vars, n_factors, seed=0, max_arity=3, dom_choices=(2, 3, 4, 5),
                 unary_noise=0.01, float_tables=True, names=True) -> FlatGraph:
    """Small heterogeneous instances for parity tests: mixed domain sizes,
    arities 1..max_arity, real-valued tables."""
    rng = np.random.default_rng(seed)
    dom_size = rng.choice(np.array(dom_choices), size=n_vars).astype(np.int32)
    var_cost = rng.uniform(0.0, unary_noise, size=int(dom_size.sum())) if unary_noise \
        else np.zeros(int(dom_size.sum()))
    rowptr, edge_var, tables, table_off = [0], [], [], [0]
    for _ in range(n_factors):
        a = int(rng.integers(1, min(max_arity, n_vars) + 1))
        sc = rng.choice(n_vars, size=a, replace=False)
        edge_var.extend(int(x) for x in sc)
        rowptr.append(len(edge_var))
        size = int(np.prod(dom_size[sc]))
        t = rng.uniform(-5, 5, size=size) if float_tables else rng.integers(0, 10, size=size).astype(float)
        tables.append(t)
        table_off.append(table_off[-1] + size)
    return _finish(dom_size, var_cost, np.array(rowptr, dtype=np.int32),
                   np.array(edge_var, dtype=np.int32), np.concatenate(tables),
                   np.array(table_off, dtype=np.int64), names)


def peav_problem(n_events, n_resources, slots=23, max_length=7, max_resources_event=5, max_value=10, seed=0):
    """The problem definition of the reference's `generate_problem_definition`
    (meetingscheduling.py:380-438) with a seeded generator: per resource the value of every slot
    if kept free (`randint(0, max_value)`), per event its length (`randint(1, max_length)`), its
    resources (`sample(resources, randint(1, max_resources_event))`) and their values
    (`randint(1, max_value)`).  -> (value_free [R, slots + 1] (column 0 unused), lengths [E],
    list of (resource ids, values) per event)."""
    rng = np.random.default_rng(seed)
    value_free = np.zeros((n_resources, slots + 1), dtype=np.int64)
    value_free[:, 1:] = rng.integers(0, max_value + 1, size=(n_resources, slots))
    lengths = rng.integers(1, max_length + 1, size=n_events)
    counts = rng.integers(1, min(max_resources_event, n_resources) + 1, size=n_events)
    events = []
    for i in range(n_events):
        res = rng.choice(n_resources, size=int(counts[i]), replace=False) if n_resources < 64 else None
        if res is None:  # (large pools: draw and re-draw the rare duplicates)
            res = rng.integers(0, n_resources, size=int(counts[i]))
            while len(set(res.tolist())) < len(res):
                res = rng.integers(0, n_resources, size=int(counts[i]))
        events.append((res.astype(np.int64), rng.integers(1, max_value + 1, size=int(counts[i]))))
    return value_free, lengths, events


def peav_like(n_events=16_700, n_resources=12_500, slots=23, max_length=7, max_resources_event=5,
              max_value=10, seed=0, penalty=None, unary_noise=0.01, names=True) -> FlatGraph:
    """The reference's PEAV meeting-scheduling DCOP (`pydcop generate meetings`,
    meetingscheduling.py:211-365), built in O(E): one variable per (resource, event) the resource
    takes part in, in the reference's order (resources, then events); the tables are the
    reference's expressions (`peav_intra_extensive_constraint_value` :540-585 with
    `resource_value_for_event` :602-640, `peav_inter_extensive_constraint` :588-599).  The defaults
    give about 50 000 variables with domains of 18..24 values and 165 000 factors.  To be solved
    with mode 'max'.  `penalty` defaults to the reference's `max_value * slots * n_resources`
    (:222)."""
    value_free, lengths, events = peav_problem(n_events, n_resources, slots, max_length,
                                               max_resources_event, max_value, seed)
    if penalty is None:
        penalty = max_value * slots * n_resources
    penalty = float(penalty)
    # variables: per resource, the events it takes part in (event order)
    per_res = [[] for _ in range(n_resources)]          # (event id, value of the resource for it)
    for ev, (res, vals) in enumerate(events):
        for r, v in zip(res.tolist(), vals.tolist()):
            per_res[r].append((ev, v))
    var_of = {}
    dom_size = []
    for r in range(n_resources):
        per_res[r].sort()
        for ev, _ in per_res[r]:
            var_of[(r, ev)] = len(dom_size)
            dom_size.append(slots - int(lengths[ev]) + 2)
    dom_size = np.array(dom_size, dtype=np.int32)
    n_vars = len(dom_size)
    csum = np.concatenate([np.zeros((n_resources, 1), dtype=np.int64), np.cumsum(value_free[:, 1:], axis=1)], axis=1)

    def utility(r, ev, val):  # resource_value_for_event for t = 0 .. D - 1 (integers)
        ln = int(lengths[ev])
        D = slots - ln + 2
        t = np.arange(1, D)
        free = csum[r, t + ln - 1] - csum[r, t - 1]
        return np.concatenate([[0], val * ln - free]).astype(np.float64)

    rowptr, edge_var, tables, table_off = [0], [], [], [0]

    def add(scope, tab):
        edge_var.extend(scope)
        rowptr.append(len(edge_var))
        tables.append(np.ascontiguousarray(tab, dtype=np.float64).reshape(-1))
        table_off.append(table_off[-1] + tab.size)

    for r in range(n_resources):
        evs = per_res[r]
        n = len(evs)
        if n == 1:
            ev, val = evs[0]
            add([var_of[(r, ev)]], utility(r, ev, val))
            continue
        us = [utility(r, ev, val) for ev, val in evs]
        for i in range(n):
            for j in range(i + 1, n):
                (e1, _), (e2, _) = evs[i], evs[j]
                l1, l2 = int(lengths[e1]), int(lengths[e2])
                t1 = np.arange(len(us[i]))[:, None]
                t2 = np.arange(len(us[j]))[None, :]
                tab = 1 / (n - 1) * (us[i][:, None] + us[j][None, :])
                clash = (t1 != 0) & (t2 != 0) & (((t1 <= t2) & (t2 <= t1 + l1 - 1)) | ((t2 <= t1) & (t1 <= t2 + l2 - 1)))
                add([var_of[(r, e1)], var_of[(r, e2)]], np.where(clash, -penalty, tab))
    for ev, (res, _) in enumerate(events):
        res = res.tolist()
        D = slots - int(lengths[ev]) + 2
        eq = np.where(np.eye(D, dtype=bool), 0.0, -penalty)
        for i in range(len(res)):
            for j in range(i + 1, len(res)):
                add([var_of[(res[i], ev)], var_of[(res[j], ev)]], eq)
    rng = np.random.default_rng(seed + 7919)
    var_cost = rng.uniform(0.0, unary_noise, size=int(dom_size.sum())) if unary_noise else np.zeros(int(dom_size.sum()))
    return _finish(dom_size, var_cost, np.array(rowptr, dtype=np.int32), np.array(edge_var, dtype=np.int32),
                   np.concatenate(tables) if tables else np.zeros(0), np.array(table_off, dtype=np.int64), names)


def secp_like(n_lights=60_000, n_models=40_000, n_rules=50_000, max_model_size=3, max_rule_size=3, seed=0,
              unary_noise=0.0, names=True, return_spec=False) -> FlatGraph:
    """The reference's smart-environment configuration problem (`pydcop generate secp`,
    pydcop/commands/generators/secp.py:127-176), built in O(E) on arrays: every variable has the five
    values `range(0, 5)` (:138).
      lights  `n_lights` variables l_i, each with a unary cost constraint `l_i * efficiency`,
              efficiency = randint(0, 90) / 100 (build_lights :302-318) -- real-valued tables of 5;
      models  `n_models` variables m_j, each with ONE constraint over 2..max_model_size lights and
              m_j: `0 if 10 * abs(m_j - (l_a * i_a + l_b * i_b ..)) < 5 else 10000`, impacts
              randint(1, 7) / 10 (build_models :201-236) -- arity 3..max_model_size + 1, integer tables of
              5^arity entries in {0, 10000} (scope order: the lights, then the model variable);
      rules   `n_rules` constraints `10 * (abs(v - target) + ..)` over 1..max_rule_size lights / models,
              targets randint(0, 4) (build_rules :239-299) -- arity 1..3, integer tables in 0..120.
    Objective min (:162).  The defaults give 100 000 variables and 150 000 factors.
    `return_spec`: also the drawn parameters (efficiencies, impacts, targets, scopes) -- what
    tests/test_secp_generator_vs_reference.py writes the reference's expression strings from."""
    rng = np.random.default_rng(seed)
    D = 5
    n_vars = n_lights + n_models
    dom_size = np.full(n_vars, D, dtype=np.int32)
    var_cost = rng.uniform(0.0, unary_noise, size=n_vars * D) if unary_noise else np.zeros(n_vars * D)
    vals = np.arange(D, dtype=np.float64)
    scopes, tabs = [], []            # per group of equal arity: (n, arity) scopes, (n, 5^arity) tables
    spec = {"efficiency": None, "models": [], "rules": []}

    # lights' efficiency costs: unary
    eff = rng.integers(0, 91, size=n_lights) / 100
    scopes.append(np.arange(n_lights, dtype=np.int64)[:, None])
    tabs.append(vals[None, :] * eff[:, None])
    spec["efficiency"] = eff

    def distinct(n, k, hi):         # n rows of k distinct integers below hi
        s = rng.integers(0, hi, size=(n, k))
        for _ in range(16):
            bad = (np.diff(np.sort(s, axis=1), axis=1) == 0).any(axis=1) if k > 1 else np.zeros(n, bool)
            if not bad.any():
                break
            s[bad] = rng.integers(0, hi, size=(int(bad.sum()), k))
        return s

    # models: by size
    size = rng.integers(2, max_model_size + 1, size=n_models)
    for k in range(2, max_model_size + 1):
        idx = np.flatnonzero(size == k)
        if idx.size == 0:
            continue
        lights = distinct(idx.size, k, n_lights)
        impact = rng.integers(1, 8, size=(idx.size, k)) / 10
        grids = np.meshgrid(*([vals] * (k + 1)), indexing="ij")          # l_1 .. l_k, m
        expr = np.zeros((idx.size,) + (D,) * (k + 1))
        for i in range(k):                                               # " l_a * i_a + l_b * i_b ..": left to right
            term = grids[i][None] * impact[:, i].reshape((-1,) + (1,) * (k + 1))
            expr = term if i == 0 else expr + term
        tab = np.where(10 * np.abs(grids[k][None] - expr) < 5, 0.0, 10000.0)
        scopes.append(np.concatenate([lights, (n_lights + idx)[:, None]], axis=1))
        tabs.append(tab.reshape(idx.size, -1))
        spec["models"].append((scopes[-1], impact))

    # rules: by (number of lights, number of models)
    rsize = rng.integers(1, min(max_rule_size, n_vars) + 1, size=n_rules)
    n_l = (rng.random(n_rules) * (rsize + 1)).astype(np.int64)           # randint(0, rule_size)
    n_l = np.minimum(n_l, rsize)
    if n_models == 0:
        n_l = rsize
    for k in range(1, max_rule_size + 1):
        for a in range(0, k + 1):
            idx = np.flatnonzero((rsize == k) & (n_l == a))
            if idx.size == 0:
                continue
            parts = []
            if a:
                parts.append(distinct(idx.size, a, n_lights))
            if k - a:
                parts.append(n_lights + distinct(idx.size, k - a, n_models))
            scope = np.concatenate(parts, axis=1)
            target = rng.integers(0, 5, size=(idx.size, k)).astype(np.float64)
            grids = np.meshgrid(*([vals] * k), indexing="ij")
            tot = np.zeros((idx.size,) + (D,) * k)
            for i in range(k):
                tot = tot + np.abs(grids[i][None] - target[:, i].reshape((-1,) + (1,) * k))
            scopes.append(scope)
            tabs.append((10 * tot).reshape(idx.size, -1))
            spec["rules"].append((scope, target))

    arity = np.concatenate([np.full(s.shape[0], s.shape[1], dtype=np.int64) for s in scopes])
    factor_rowptr = np.concatenate([[0], np.cumsum(arity)]).astype(np.int32)
    edge_var = np.concatenate([s.reshape(-1) for s in scopes]).astype(np.int32)
    sizes = np.concatenate([np.full(t.shape[0], t.shape[1], dtype=np.int64) for t in tabs])
    table_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    tables = np.concatenate([t.reshape(-1) for t in tabs]).astype(np.float64)
    g = _finish(dom_size, var_cost, factor_rowptr, edge_var, tables, table_off, names)
    return (g, spec) if return_spec else g
