"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Runs the *reference's own* synchronous Max-Sum computations
(`/root/reference/pydcop/algorithms/maxsum.py`) deterministically for an exact
number of cycles, without agents, threads or the orchestrator.  Used to

  * pin `oracle/maxsum_oracle.c` against the real reference (tests run here,
    in the build container, where `/root/reference` exists), and
  * generate the committed golden vectors under `tests/golden/`
    (`oracle/make_golden.py`).

`/root/reference` does not exist on the GPU box; what travels there is the C
restatement, the golden fixtures and -- git-ignored, packed by
`oracle/stage_reference.py` at build time -- an archive of the reference itself
(`oracle/_ref/`), which `stage_reference.locate()` unpacks outside the repository.

Shims (reference untouched, all in-process):
  1. `collections.Iterable/Mapping/...` aliases (pydcop/dcop/yamldcop.py:32,
     pydcop/dcop/dcop.py:238 use the pre-3.10 names),
  2. stub modules for the absent third-party deps `websocket_server`
     (pydcop/infrastructure/ui.py:36) and `pulp` (pydcop/distribution/ilp_*.py),
  3. `ndarray.itemset` was removed in numpy 2 (pydcop/dcop/relations.py:857).
"""
import collections
import collections.abc
import os
import sys
import types
from collections import deque

try:
    from oracle import stage_reference as _stage
except ImportError:  # run as a script from inside oracle/
    import stage_reference as _stage

REFERENCE_ROOT = _stage.locate() or "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pydcop"))


class _Permissive(types.ModuleType):
    """A module whose every attribute is a do-nothing class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        stub = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, stub)
        return stub


def install_shims():
    """Make `import pydcop` work on python 3.10 / numpy 2 without touching it."""
    sys.dont_write_bytecode = True  # never write __pycache__ into the (read-only) reference tree
    for n in ("Iterable", "Mapping", "Sequence", "Callable", "Sized",
              "MutableMapping", "Hashable", "Set"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    for mod in ("websocket_server", "websocket_server.websocket_server",
                "pulp", "pulp.constants", "pulp.pulp", "pulp.solvers"):
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                sys.modules[mod] = _Permissive(mod)
    if reference_available() and REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if reference_available():
        _patch_itemset()


def _patch_itemset():
    # pydcop/dcop/relations.py:857 calls ndarray.itemset (gone in numpy 2).
    from pydcop.dcop import relations as R

    if getattr(R.NAryMatrixRelation, "_graft_patched", False):
        return
    import numpy as np

    if hasattr(np.zeros(1), "itemset"):  # numpy 2 keeps a class attribute that raises on instances
        return

    def set_value_for_assignment(self, var_values, rel_value):
        # numpy-2 stand-in: same contract (a NEW relation with one entry changed), the
        # entry addressed through the relation's own _slice_matrix
        names = [v.name for v in self._variables]
        if isinstance(var_values, dict):
            var_values = [var_values[n] for n in names]
        elif not isinstance(var_values, list):
            raise ValueError("Could not set value, must be list or dict")
        _, where = self._slice_matrix(names, var_values)
        changed = np.array(self._m, copy=True)
        changed[where] = rel_value
        return R.NAryMatrixRelation(self._variables, changed, name=self.name)

    R.NAryMatrixRelation.set_value_for_assignment = set_value_for_assignment
    R.NAryMatrixRelation._graft_patched = True


def run_reference_maxsum(dcop, cycles, params=None, cg=None, return_comps=False):
    """Run the reference's MaxSum{Factor,Variable}Computation objects for exactly
    `cycles` calls of `on_new_cycle` each.

    Every computation's `message_sender` appends to one FIFO; messages whose
    `cycle_id >= cycles` are dropped so that no computation can switch to cycle
    `cycles+1`.  Returns ({var: value}, {var: cost}).
    """
    install_shims()
    from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
    from pydcop.computations_graph import factor_graph

    p = {"noise": 0}
    p.update(params or {})
    algo = AlgorithmDef.build_with_default_param("maxsum", p, mode=dcop.objective)
    if cg is None:
        cg = factor_graph.build_computation_graph(dcop)
    module = load_algorithm_module("maxsum")
    comps = {}
    q = deque()
    import logging
    logging.disable(logging.CRITICAL)
    for node in cg.nodes:
        c = module.build_computation(ComputationDef(node, algo))
        comps[node.name] = c

    def make_sender():
        def sender(src, dest, msg, prio=None, on_error=None):
            q.append((src, dest, msg))
        return sender

    for c in comps.values():
        c.message_sender = make_sender()
    try:
        for c in comps.values():
            c.start()
        while q:
            s, d, m = q.popleft()
            if m.cycle_id < cycles:
                comps[d].on_message(s, m, 0.0)
    finally:
        logging.disable(logging.NOTSET)
    values = {v: comps[v].current_value for v in dcop.variables}
    costs = {v: comps[v].current_cost for v in dcop.variables}
    if return_comps:
        return values, costs, comps
    return values, costs


class ReferenceMaxSumRun:
    """The reference's synchronous Max-Sum computations under the FIFO harness of
    `run_reference_maxsum`, RESUMABLE: `run_to(T)` delivers every queued message whose cycle_id
    is below T (each computation then has executed exactly T calls of on_new_cycle) and parks the
    rest, so that something can happen to the computations between two cycles -- what
    pydcop/algorithms/maxsum_dynamic.py does to a factor:

      change_factor_function(name, relation)
          runs the reference's OWN `DynamicFunctionFactorComputation.change_factor_function`
          (maxsum_dynamic.py:80-104: the two dimension checks, then `self.factor = fn`; the
          re-emission of messages is commented out there, "FIXME") on the factor's computation.
          The reference defines that class over the ASYNCHRONOUS factor computation (`from
          pydcop.algorithms.amaxsum import MaxSumFactorComputation`, maxsum_dynamic.py:35); the
          method body touches nothing but `self.factor`, which the synchronous computation reads
          in the same way (maxsum.py:345: `factor_costs_for_var(self.factor, ...)`), so it is
          applied to the synchronous computation object unchanged.
      slice_external(relation, values)
          `relation.slice(values)` (relations.py:760-810) -- what
          FactorWithReadOnlyVariableComputation._on_new_var_value_msg (maxsum_dynamic.py:159-176)
          computes before it calls change_factor_function.  (That class itself cannot be
          constructed: its __init__ calls `super().__init__(relation, name=..., msg_sender=...)`
          on a parent that takes a single comp_def, maxsum_dynamic.py:124-143 -- a TypeError,
          checked by tests/test_dynamic_vs_reference.py.)"""

    def __init__(self, dcop, params=None, cg=None):
        install_shims()
        from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
        from pydcop.computations_graph import factor_graph
        import logging
        p = {"noise": 0}
        p.update(params or {})
        algo = AlgorithmDef.build_with_default_param("maxsum", p, mode=dcop.objective)
        self.dcop = dcop
        self.cg = cg if cg is not None else factor_graph.build_computation_graph(dcop)
        module = load_algorithm_module("maxsum")
        self.q = deque()
        self.comps = {}
        logging.disable(logging.CRITICAL)
        try:
            for node in self.cg.nodes:
                c = module.build_computation(ComputationDef(node, algo))
                c.message_sender = lambda src, dest, msg, prio=None, on_error=None: self.q.append((src, dest, msg))
                self.comps[node.name] = c
            for c in self.comps.values():
                c.start()
        finally:
            logging.disable(logging.NOTSET)
        self.cycles = 0

    def run_to(self, T):
        import logging
        logging.disable(logging.CRITICAL)
        try:
            parked = deque()
            while self.q:
                s, d, m = self.q.popleft()
                if m.cycle_id < T:
                    self.comps[d].on_message(s, m, 0.0)
                else:
                    parked.append((s, d, m))
            self.q = parked
        finally:
            logging.disable(logging.NOTSET)
        self.cycles = T

    def change_factor_function(self, name, relation):
        from pydcop.algorithms.maxsum_dynamic import DynamicFunctionFactorComputation
        DynamicFunctionFactorComputation.change_factor_function(self.comps[name], relation)

    @staticmethod
    def slice_external(relation, values):
        return relation.slice(values)

    def values(self):
        return ({v: self.comps[v].current_value for v in self.dcop.variables},
                {v: self.comps[v].current_cost for v in self.dcop.variables})


class ReferenceDynamicAMaxSumRun:
    """The reference's ASYNCHRONOUS Max-Sum computations under the FIFO delivery of
    `run_reference_amaxsum`, resumable: `run(G)` delivers the generations below G; between two calls
    `change_factor_function(name, relation)` runs the reference's own
    `DynamicFunctionFactorComputation.change_factor_function` (maxsum_dynamic.py:80-104) on the
    factor's computation -- an object of the very class the dynamic one derives from (`from
    pydcop.algorithms.amaxsum import MaxSumFactorComputation`, maxsum_dynamic.py:35).

    `dynamic_class=True` builds `DynamicFunctionFactorComputation` objects themselves instead: they
    construct, but cannot handle a single message -- the `@register("max_sum")` handler of the parent
    is not in the subclass's handler table (KeyError in computations.py:509; "does not work since the
    refactoring", maxsum_dynamic.py:60).  tests/test_dynamic_vs_reference.py asserts that, which is
    why the method is exercised on the parent class's objects."""

    def __init__(self, dcop, params=None, cg=None, dynamic_class=False):
        install_shims()
        from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
        from pydcop.algorithms.maxsum_dynamic import DynamicFunctionFactorComputation
        from pydcop.computations_graph import factor_graph
        import logging
        p = {"noise": 0}
        p.update(params or {})
        algo = AlgorithmDef.build_with_default_param("amaxsum", p, mode=dcop.objective)
        self.dcop = dcop
        self.cg = cg if cg is not None else factor_graph.build_computation_graph(dcop)
        module = load_algorithm_module("amaxsum")
        self.q = deque()
        self.comps = {}
        self.handling = -1
        self.delivered = 0
        logging.disable(logging.CRITICAL)
        try:
            for node in self.cg.nodes:
                cd = ComputationDef(node, algo)
                c = (DynamicFunctionFactorComputation(comp_def=cd)
                     if dynamic_class and node.type == "FactorComputation" else module.build_computation(cd))
                c.message_sender = self._send
                self.comps[node.name] = c
            for c in self.comps.values():
                c.start()
        finally:
            logging.disable(logging.NOTSET)

    def _send(self, src, dest, msg, prio=None, on_error=None):
        self.q.append((src, dest, msg, self.handling + 1))

    def run(self, max_generations):
        import logging
        logging.disable(logging.CRITICAL)
        try:
            while self.q and self.q[0][3] < max_generations:
                s, d, m, g = self.q.popleft()
                self.handling = g
                self.comps[d].on_message(s, m, 0.0)
                self.delivered += 1
        finally:
            logging.disable(logging.NOTSET)
        return self.delivered

    def change_factor_function(self, name, relation):
        from pydcop.algorithms.maxsum_dynamic import DynamicFunctionFactorComputation
        DynamicFunctionFactorComputation.change_factor_function(self.comps[name], relation)

    def values(self):
        return ({v: self.comps[v].current_value for v in self.dcop.variables},
                {v: self.comps[v].current_cost for v in self.dcop.variables})


def reference_message_state(comps, graph):
    """What the reference's synchronous computations HOLD, laid out like the flat message
    buffers (graph.msg_off per factor-major edge), after `run_reference_maxsum(...,
    return_comps=True)`:

      sent_f2v / count_f2v   the factor's `_prev_messages[variable]` = (last SENT message, count)
                             (maxsum.py:303, 343-377)
      sent_v2f / count_v2f   the variable's `_prev_messages[factor]` (maxsum.py:474, 529-564)
      held_v2f               the factor's `_costs[variable]`: what it last RECEIVED (maxsum.py:294, 342)
      held_f2v               the variable's `costs[factor]` (maxsum.py:466, 528)

    NaN = nothing there (no message sent / received yet); `has_<key>` says the same as a boolean
    mask, for instances whose messages hold real NaNs (hard constraints: inf - inf)."""
    import numpy as np
    g = graph
    nm, ne = int(g.msg_off[-1]), g.n_edges
    names = g.var_names
    fnames = g.factor_names or [f"c{i}" for i in range(g.n_factors)]
    keys = ("sent_f2v", "sent_v2f", "held_v2f", "held_f2v")
    out = {k: np.full(nm, np.nan) for k in keys}
    out.update({"has_" + k: np.zeros(nm, dtype=bool) for k in keys})
    out["count_f2v"] = np.zeros(ne, dtype=np.uint8)
    out["count_v2f"] = np.zeros(ne, dtype=np.uint8)

    def put(key, e, costs, domain):
        o = int(g.msg_off[e])
        for d, val in enumerate(domain):
            out[key][o + d] = costs[val]
            out["has_" + key][o + d] = True

    for f in range(g.n_factors):
        fc = comps[fnames[f]]
        for e in range(int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])):
            v = int(g.edge_var[e])
            vc = comps[names[v]]
            domain = list(vc.variable.domain)
            msg, cnt = fc._prev_messages[names[v]]
            out["count_f2v"][e] = cnt
            if msg is not None:
                put("sent_f2v", e, msg, domain)
            if names[v] in fc._costs:
                put("held_v2f", e, fc._costs[names[v]], domain)
            msg, cnt = vc._prev_messages[fnames[f]]
            out["count_v2f"][e] = cnt
            if msg is not None:
                put("sent_v2f", e, msg, domain)
            if fnames[f] in vc.costs:
                put("held_f2v", e, vc.costs[fnames[f]], domain)
    return out


def run_reference_amaxsum(dcop, max_generations=-1, params=None, cg=None, max_messages=None):
    """The reference's ASYNCHRONOUS Max-Sum computations (pydcop/algorithms/amaxsum.py) under
    FIFO delivery: every computation started in graph order, ONE queue, messages handled first
    in first out -- the delivery order oracle/amaxsum_oracle.c restates.  Generation 0 = the
    start messages; a message sent while handling one of generation g belongs to g + 1; only
    generations < max_generations are delivered (all of them: -1; the run ends by itself when the
    send rule of amaxsum.py:222-244 has silenced every edge).

    Returns ({var: value}, {var: cost}, info) with info = {"delivered", "generation_sizes",
    "comps"}."""
    install_shims()
    from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
    from pydcop.computations_graph import factor_graph

    p = {"noise": 0}
    p.update(params or {})
    algo = AlgorithmDef.build_with_default_param("amaxsum", p, mode=dcop.objective)
    if cg is None:
        cg = factor_graph.build_computation_graph(dcop)
    module = load_algorithm_module("amaxsum")
    import logging
    logging.disable(logging.CRITICAL)
    comps = {}
    q = deque()
    sizes = {}
    handling = [-1]

    def sender(src, dest, msg, prio=None, on_error=None):
        g = handling[0] + 1
        sizes[g] = sizes.get(g, 0) + 1
        q.append((src, dest, msg, g))

    delivered = 0
    try:
        for node in cg.nodes:
            comps[node.name] = module.build_computation(ComputationDef(node, algo))
            comps[node.name].message_sender = sender
        for c in comps.values():
            c.start()
        while q:
            if max_generations >= 0 and q[0][3] >= max_generations:
                break
            if max_messages is not None and delivered >= max_messages:
                break
            s, d, m, g = q.popleft()
            handling[0] = g
            comps[d].on_message(s, m, 0.0)
            delivered += 1
    finally:
        logging.disable(logging.NOTSET)
    values = {v: comps[v].current_value for v in dcop.variables}
    costs = {v: comps[v].current_cost for v in dcop.variables}
    info = {"delivered": delivered, "generation_sizes": [sizes[k] for k in sorted(sizes)], "comps": comps,
            "pending": len(q)}
    return values, costs, info


def run_reference_mgm(dcop, rounds, cg=None):
    """The reference's own MgmComputation objects (pydcop/algorithms/mgm.py) for exactly `rounds`
    rounds of (values, gains, decision): stop_cycle = rounds + 1 (mgm.py:407-411), FIFO delivery
    (MGM parks early messages, so any order gives the same result).  The reference's three draws
    from the unseeded `random` module are made deterministic: random.choice -> the first element
    (initial value, choice among equally good values), random.random -> 0.

    Returns ({var: value}, {var: cost}, comps)."""
    install_shims()
    from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
    from pydcop.computations_graph import constraints_hypergraph as chg
    import pydcop.algorithms.mgm as mgm
    import logging

    class _FirstChoice:
        def __getattr__(self, name):
            import random as _r
            return getattr(_r, name)

        @staticmethod
        def choice(seq):
            return seq[0]

        @staticmethod
        def random():
            return 0.0

    saved = mgm.random
    mgm.random = _FirstChoice()
    logging.disable(logging.CRITICAL)
    try:
        if cg is None:
            cg = chg.build_computation_graph(dcop)
        algo = AlgorithmDef.build_with_default_param("mgm", {"stop_cycle": rounds + 1}, mode=dcop.objective)
        module = load_algorithm_module("mgm")
        comps, q = {}, deque()

        def sender(src, dest, msg, prio=None, on_error=None):
            q.append((src, dest, msg))

        for node in cg.nodes:
            c = module.build_computation(ComputationDef(node, algo))
            c.message_sender = sender
            c._on_finished = lambda *a, **k: None   # (no agent to tell)
            comps[node.name] = c
        for c in comps.values():
            c.start()
        while q:
            s, d, m = q.popleft()
            comps[d].on_message(s, m, 0.0)
    finally:
        mgm.random = saved
        logging.disable(logging.NOTSET)
    values = {v: comps[v].current_value for v in dcop.variables}
    costs = {v: comps[v].current_cost for v in dcop.variables}
    return values, costs, comps


_M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def dsa_uniform(seed, variable, cycle, draw):
    """The counter-based generator of oracle/dsa_oracle.c (and pydcop_amd/csrc/dsa.hip), bit for bit."""
    z = (seed + 0x9E3779B97F4A7C15 * (variable + 1)) & _M64
    z = (_mix64(z) + 0x9E3779B97F4A7C15 * (cycle + 1)) & _M64
    z = (_mix64(z) + draw) & _M64
    return (_mix64(z) >> 11) * (1.0 / 9007199254740992.0)


def run_reference_dsa(dcop, cycles, variant="B", probability=0.7, p_mode="fixed", seed=0, var_index=None):
    """The reference's own DsaComputation objects (pydcop/algorithms/dsa.py) for exactly `cycles`
    evaluations each (stop_cycle = cycles), FIFO delivery (DSA parks early values, dsa.py:300-317,
    so any order gives the same result).  The reference's draws from the unseeded `random`
    module are replaced -- for the duration of the run -- by dsa_uniform keyed on (variable,
    cycle, draw), see oracle/dsa_oracle.c.  Returns ({var: value}, {var: cost}, comps)."""
    install_shims()
    from pydcop.algorithms import AlgorithmDef, ComputationDef, load_algorithm_module
    from pydcop.computations_graph import constraints_hypergraph as chg
    import pydcop.algorithms.dsa as dsa
    import pydcop.infrastructure.computations as comps_mod
    import logging

    names = sorted(dcop.variables) if var_index is None else None
    index = var_index or {n: i for i, n in enumerate(names)}
    ctx = {"comp": None, "start": False}

    class _Keyed:
        def __getattr__(self, name):
            import random as _r
            return getattr(_r, name)

        @staticmethod
        def random():
            c = ctx["comp"]
            return dsa_uniform(seed, index[c.name], c.cycle_count + 1, 1)

        @staticmethod
        def choice(seq):
            c = ctx["comp"]
            if ctx["start"]:
                u = dsa_uniform(seed, index[c.name], 0, 0)
            else:
                u = dsa_uniform(seed, index[c.name], c.cycle_count + 1, 2)
            return seq[int(u * len(seq))]

    saved = (dsa.random, comps_mod.random)
    dsa.random = comps_mod.random = _Keyed()
    logging.disable(logging.CRITICAL)
    try:
        cg = chg.build_computation_graph(dcop)
        algo = AlgorithmDef.build_with_default_param(
            "dsa", {"stop_cycle": cycles, "variant": variant, "probability": probability, "p_mode": p_mode},
            mode=dcop.objective)
        module = load_algorithm_module("dsa")
        comps, q = {}, deque()

        def sender(src, dest, msg, prio=None, on_error=None):
            q.append((src, dest, msg))

        for node in cg.nodes:
            c = module.build_computation(ComputationDef(node, algo))
            c.message_sender = sender
            c._on_finished = lambda *a, **k: None
            comps[node.name] = c
        ctx["start"] = True
        for c in comps.values():
            ctx["comp"] = c
            c.start()
        ctx["start"] = False
        while q and cycles > 0:
            s, d, m = q.popleft()
            ctx["comp"] = comps[d]
            comps[d].on_message(s, m, 0.0)
    finally:
        dsa.random, comps_mod.random = saved
        logging.disable(logging.NOTSET)
    values = {v: comps[v].current_value for v in dcop.variables}
    costs = {v: comps[v].current_cost for v in dcop.variables}
    return values, costs, comps


def flat_to_dcop(graph, mode="min", name="flat"):
    """Build reference objects (DCOP + ComputationsFactorGraph) from a FlatGraph:
    VariableWithCostDict variables (pydcop/dcop/objects.py:410) and extensional
    NAryMatrixRelation constraints (pydcop/dcop/relations.py:672).  The factor
    graph is assembled in O(E) instead of through the quadratic
    build_computation_graph (pydcop/computations_graph/factor_graph.py:245),
    with the same node/link structure."""
    install_shims()
    import numpy as np
    from pydcop.computations_graph.factor_graph import (
        ComputationsFactorGraph, FactorComputationNode, VariableComputationNode)
    from pydcop.dcop.dcop import DCOP
    from pydcop.dcop.objects import Domain, VariableWithCostDict
    from pydcop.dcop.relations import NAryMatrixRelation

    g = graph
    names = g.var_names or [f"v{i}" for i in range(g.n_vars)]
    fnames = g.factor_names or [f"c{i}" for i in range(g.n_factors)]
    cost_off = g.cost_off
    doms = {}
    variables = []
    dcop = DCOP(name, objective=mode)
    for i in range(g.n_vars):
        D = int(g.dom_size[i])
        values = list(g.domains[i]) if g.domains else list(range(D))
        key = tuple(values)
        if key not in doms:
            doms[key] = Domain(f"d{len(doms)}", "d", values)
        costs = {values[d]: float(g.var_cost[cost_off[i] + d]) for d in range(D)}
        init = None
        if g.init_idx is not None and g.init_idx[i] >= 0:
            init = values[int(g.init_idx[i])]
        v = VariableWithCostDict(names[i], doms[key], costs, initial_value=init)
        variables.append(v)
        dcop.add_variable(v)
    factor_nodes = []
    for f in range(g.n_factors):
        e0, e1 = int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])
        scope = [variables[int(x)] for x in g.edge_var[e0:e1]]
        shape = tuple(len(v.domain) for v in scope)
        m = np.array(g.tables[g.table_off[f]:g.table_off[f + 1]]).reshape(shape)
        c = NAryMatrixRelation(scope, m, name=fnames[f])
        dcop.add_constraint(c)
        factor_nodes.append(FactorComputationNode(c))
    edge_factor = np.repeat(np.arange(g.n_factors), np.diff(g.factor_rowptr))
    var_nodes = []
    for i in range(g.n_vars):
        k0, k1 = int(g.var_rowptr[i]), int(g.var_rowptr[i + 1])
        cnames = [fnames[int(edge_factor[e])] for e in g.var_edges[k0:k1]]
        var_nodes.append(VariableComputationNode(variables[i], cnames))
    cg = ComputationsFactorGraph(var_nodes, factor_nodes)
    return dcop, cg
