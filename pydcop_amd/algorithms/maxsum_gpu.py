"""maxsum_gpu -- synchronous Max-Sum as one batched GPU sweep, behind pyDCOP's
algorithm-module contract (docs/implementation/algorithms.rst; enforced by
pydcop/algorithms/__init__.py:527-566).

Drop-in for `pydcop.algorithms.maxsum`: same GRAPH_TYPE, the same five
parameters with the same defaults (pydcop/algorithms/maxsum.py:212-220), the
same computation_memory / communication_load formulas, and a
`build_computation(comp_def)` factory.  The reference hands the plugin ONE node
at a time; the computations returned here are thin proxies that register their
ComputationDef with a process-global session.  When the first proxy is started
the session checks that the registered set is closed (every neighbour is
registered), compiles the factor graph to flat arrays (pydcop_amd.compile),
creates one `MaxSumEngine` (HIP, MI355X) and sweeps the whole graph; every
variable proxy then reports its value through `value_selection` on its own
agent's thread, which is all the unmodified orchestrator needs to produce the
usual result JSON (pydcop/infrastructure/orchestrator.py:1215-1274).

Extra parameters
  stop_cycle  int, default 0   >0: run exactly that many cycles then call
                               finished() (status FINISHED); 0: keep sweeping in
                               chunks until the orchestrator's timeout, like the
                               reference (which never terminates, maxsum.py:62)
  precision   f64 (reference arithmetic, default) | f32
  seed        seed of the noise draw (the reference's noise is unseeded,
              pydcop/dcop/objects.py:566-567)
  chunk       cycles per device run between two reports when stop_cycle == 0
  devices     number of GPUs of this node to partition the factor graph over (default 1):
              k > 1 runs `pydcop_amd.sharded.LocalShardedMaxSum` -- one engine per GPU,
              boundary V->F messages exchanged once per cycle by the engines' own RCCL
              all-to-all -- and gives the same result as one GPU

Only thread mode with all computations in one process can be served (the whole
graph must be visible to one engine); anything else raises ComputationException.
"""
import threading
from typing import Dict, Optional, Union

import numpy as np

from pydcop.algorithms import AlgoParameterDef, ComputationDef
from pydcop.computations_graph.factor_graph import (FactorComputationNode,
                                                     VariableComputationNode)
from pydcop.infrastructure.computations import (ComputationException, DcopComputation,
                                                VariableComputation)

from pydcop_amd.compile import compile_nodes
from pydcop_amd.graph import Params

GRAPH_TYPE = "factor_graph"

HEADER_SIZE = 0
UNIT_SIZE = 1
FACTOR_UNIT_SIZE = 1
VARIABLE_UNIT_SIZE = 1

algo_params = [
    AlgoParameterDef("damping", "float", None, 0.5),
    AlgoParameterDef("damping_nodes", "str", ["vars", "factors", "both", "none"], "both"),
    AlgoParameterDef("stability", "float", None, 0.1),
    AlgoParameterDef("noise", "float", None, 0.01),
    AlgoParameterDef("start_messages", "str", ["leafs", "leafs_vars", "all"], "leafs"),
    AlgoParameterDef("stop_cycle", "int", None, 0),
    AlgoParameterDef("precision", "str", ["f64", "f32"], "f64"),
    AlgoParameterDef("seed", "int", None, 0),
    AlgoParameterDef("chunk", "int", None, 10),
    AlgoParameterDef("devices", "int", None, 1),
]

def computation_memory(computation: Union[FactorComputationNode, VariableComputationNode]) -> float:
    """Same footprint model as pydcop/algorithms/maxsum.py:127-171."""
    if isinstance(computation, FactorComputationNode):
        return sum(len(v.domain) * FACTOR_UNIT_SIZE for v in computation.variables)
    if isinstance(computation, VariableComputationNode):
        return len(list(computation.links)) * len(computation.variable.domain) * VARIABLE_UNIT_SIZE
    raise ValueError(
        "Invalid computation node type {}, maxsum_gpu only defines VariableComputationNode "
        "and FactorComputationNode".format(computation))


def communication_load(src: Union[FactorComputationNode, VariableComputationNode],
                       target: str) -> float:
    """Same message-size model as pydcop/algorithms/maxsum.py:174-209."""
    if isinstance(src, VariableComputationNode):
        return UNIT_SIZE * len(src.variable.domain) + HEADER_SIZE
    if isinstance(src, FactorComputationNode):
        for v in src.variables:
            if v.name == target:
                return UNIT_SIZE * len(v.domain) + HEADER_SIZE
        raise ValueError("Could not find variable {} in constraint of factor {}".format(target, src))
    raise ValueError("maxsum_gpu communication_load only supports VariableComputationNode and "
                     "FactorComputationNode, invalid computation: " + str(src))


class _Session:
    """All the computations of one solve: collects ComputationDefs, owns the engine."""

    def __init__(self):
        self.lock = threading.RLock()
        self.comp_defs: Dict[str, ComputationDef] = {}
        self.engine = None
        self.graph = None
        self.var_index: Dict[str, int] = {}
        # (idx, belief, cycles, generation) of the last fetch: replaced as ONE tuple, so a
        # proxy on another agent thread never pairs a value with another generation's belief
        self.snapshot = (None, None, 0, 0)
        self.started = set()
        self.done = False
        self.stopped = False
        self.error: Optional[Exception] = None

    def register(self, comp_def: ComputationDef):
        with self.lock:
            self.comp_defs[comp_def.node.name] = comp_def

    # -- solve ------------------------------------------------------------------
    def _open(self):
        """Compile the registered graph and create the engine (cycle 0 runs there)."""
        from pydcop_amd.engine import MaxSumEngine
        missing = sorted({n for cd in self.comp_defs.values() for n in cd.node.neighbors}
                         - set(self.comp_defs))
        if missing:
            raise ComputationException(
                "maxsum_gpu needs every computation of the factor graph in one process "
                "(thread mode); not deployed here: " + ", ".join(missing[:8]))
        any_def = next(iter(self.comp_defs.values()))
        algo = any_def.algo
        p = algo.params
        self.graph = self._compile_graph(p)
        self.var_index = {n: i for i, n in enumerate(self.graph.var_names)}
        self.engine = self._make_engine(self._engine_params(algo, p), p)
        self.stop_cycle = self._cycles_to_run(p)
        self.chunk = max(1, int(p["chunk"]))
        self._fetch()

    ALGO = "maxsum_gpu"

    # -- what a session of another algorithm overrides (amaxsum_gpu, mgm_gpu) ----------------
    def _compile_graph(self, p):
        var_nodes = [cd.node for cd in self.comp_defs.values() if cd.node.type == "VariableComputation"]
        fac_nodes = [cd.node for cd in self.comp_defs.values() if cd.node.type == "FactorComputation"]
        var_nodes.sort(key=lambda n: n.name)
        fac_nodes.sort(key=lambda n: n.name)
        return compile_nodes(var_nodes, fac_nodes, noise=float(p["noise"]),
                             rng=np.random.default_rng(int(p["seed"])))

    def _engine_params(self, algo, p):
        return Params(mode=algo.mode, damping=float(p["damping"]),
                      damping_nodes=p["damping_nodes"], stability=float(p["stability"]),
                      start_messages=p["start_messages"], dtype=p["precision"])

    def _cycles_to_run(self, p) -> int:
        return int(p["stop_cycle"])

    def _make_engine(self, params, p):
        """The engine of this kind of session (amaxsum_gpu overrides it)."""
        n_dev = int(p.get("devices", 1) or 1)
        if n_dev > 1:
            from pydcop_amd.engine import device_count
            from pydcop_amd.sharded import LocalShardedMaxSum
            if device_count() < n_dev:
                raise ComputationException(
                    f"maxsum_gpu: devices:{n_dev} asked, {device_count()} GPU(s) visible")
            return LocalShardedMaxSum(self.graph, params, list(range(n_dev)))
        # (DynamicMaxSum = a MaxSumEngine that survives scope changes of its factors)
        from pydcop_amd.dynamic import DynamicMaxSum
        return DynamicMaxSum(self.graph, params)

    @property
    def cycles(self) -> int:
        return self.snapshot[2]

    @property
    def generation(self) -> int:
        return self.snapshot[3]

    def _fetch(self):
        idx, belief = self.engine.assignment()
        self.snapshot = (idx, belief, self.engine.cycle_count, self.snapshot[3] + 1)

    def advance(self):
        """Called from any proxy's agent thread: make progress if nobody else is."""
        if not self.lock.acquire(blocking=False):
            return
        try:
            if self.error or self.done or self.stopped:
                return
            if self.engine is None:
                self._open()
                if self.stop_cycle > 0:
                    self.engine.run(self.stop_cycle)
                    self._fetch()
                    self.done = True
                elif getattr(self.engine, "quiescent", False):
                    # amaxsum_gpu with nothing to deliver at all (e.g. `start_messages: leafs` on a graph
                    # without leaves): the run ends FINISHED with the initial values, where the
                    # reference sits until its TIMEOUT -- say so, it is not a converged run
                    import logging
                    logging.getLogger("pydcop.algo.amaxsum_gpu").warning(
                        "amaxsum_gpu: no start message (start_messages=%s): finished with the initial values, "
                        "0 generations", next(iter(self.comp_defs.values())).algo.params.get("start_messages"))
                    self.done = True
                return
            if self.stop_cycle == 0:
                self.engine.run(self.chunk)
                self._fetch()
                if getattr(self.engine, "quiescent", False):  # (amaxsum: no message left)
                    self.done = True
        except Exception as e:  # surfaced by every proxy
            self.error = e
            raise
        finally:
            self.lock.release()

    def wait_open(self):
        with self.lock:
            if self.error:
                raise self.error
            if self.engine is None and not self.stopped:
                self.advance()
            if self.error:
                raise self.error

    def update_factor(self, name, old, fn):
        """Tensorise `fn` and hand it to the engine: same variables -> the table is swapped in
        place; other variables -> the scope change of maxsum_dynamic.py:234-271 (re-layout, the
        messages of the surviving edges carried over; one GPU only)."""
        from pydcop_amd.compile import tensorise_constraint
        with self.lock:
            if self.engine is None:
                raise ComputationException("maxsum_gpu: the engine is not running")
            t = tensorise_constraint(fn)
            f = self.graph.factor_names.index(name)
            same = {v.name for v in old.dimensions} == {v.name for v in fn.dimensions}
            if hasattr(self.engine, "change_factor_function"):
                unknown = [v.name for v in fn.dimensions if v.name not in self.var_index]
                if unknown:
                    raise ValueError("maxsum_gpu: the new function of {} depends on variables without a "
                                     "computation: {}".format(name, ", ".join(unknown)))
                self.engine.change_factor_function(f, t, scope=[self.var_index[v.name] for v in fn.dimensions])
                self.graph = self.engine.graph
            elif same:
                src = [v.name for v in fn.dimensions]
                t = np.transpose(t, [src.index(v.name) for v in old.dimensions])
                self.engine.update_factor_table(f, t)
            else:
                raise ValueError("maxsum_gpu: changing the scope of a factor needs devices:1")

    def value_of(self, name):
        idx, belief, _, _ = self.snapshot
        i = self.var_index[name]
        return self.graph.domains[i][int(idx[i])], float(belief[i])

    def proxy_started(self, name):
        with self.lock:
            self.started.add(name)

    def proxy_stopped(self, name):
        """One computation stopped (end of the run, but also a scenario event or an agent
        removal in the middle of it): the others keep their engine.  It is closed when the
        last started proxy has stopped."""
        with self.lock:
            self.started.discard(name)
            if self.started:
                return
            self.stopped = True
            if self.engine is not None:
                self.engine.close()
                self.engine = None


_registry_lock = threading.Lock()
_current: Dict[str, _Session] = {}
SESSION_CLASSES = {"maxsum_gpu": _Session}   # algorithm name -> session class (amaxsum_gpu adds its own)


def _session_for(comp_def: ComputationDef) -> _Session:
    """The open session of the computation's algorithm, or a new one when the previous solve is
    over / already holds a computation of that name (a new run in the same process)."""
    algo = comp_def.algo.algo
    with _registry_lock:
        s = _current.get(algo)
        if (s is None or s.stopped or s.engine is not None or s.error is not None
                or comp_def.node.name in s.comp_defs):
            s = _current[algo] = SESSION_CLASSES[algo]()
        s.register(comp_def)
        return s


class _ProxyMixin:
    """Behaviour shared by the factor and variable proxies."""

    POLL_PERIOD = 0.02

    def _init_proxy(self, comp_def):
        assert comp_def.algo.algo in SESSION_CLASSES
        self._session = _session_for(comp_def)
        self._seen_generation = 0
        self._poll_handle = None
        self._reported_done = False

    @property
    def cycle_count(self):
        return self._session.cycles

    def footprint(self) -> float:
        return computation_memory(self.computation_def.node)

    def on_start(self):
        s = self._session
        s.proxy_started(self.name)
        s.wait_open()            # first proxy started compiles + creates the engine
        self._report()
        if not self._reported_done:
            self._poll_handle = self.add_periodic_action(self.POLL_PERIOD, self._tick)

    def _tick(self):
        s = self._session
        if not s.done:
            s.advance()
        if s.error:
            raise s.error
        self._report()

    def _report(self):
        s = self._session
        if s.generation != self._seen_generation:
            self._seen_generation = s.generation
            # cycle statistics (`--collect_on cycle_change`, computations.py:915-928): one event
            # per report, carrying the engine's cycle count (`chunk:1` gives one per cycle)
            self.new_cycle()
            self._publish()
        if s.done and not self._reported_done:
            self._reported_done = True
            if self._poll_handle is not None:
                self.remove_periodic_action(self._poll_handle)
                self._poll_handle = None
            self.finished()
            self.stop()

    def _publish(self):
        pass

    def on_stop(self):
        self._session.proxy_stopped(self.name)


class MaxSumGpuFactorComputation(_ProxyMixin, DcopComputation):
    """Stands for a MaxSumFactorComputation (pydcop/algorithms/maxsum.py:279): holds
    no state, its messages are computed by the factor side of the GPU sweep."""

    def __init__(self, comp_def: ComputationDef):
        super().__init__(comp_def.node.factor.name, comp_def)
        self._init_proxy(comp_def)

    def change_factor_function(self, fn):
        """New cost function: over the same variables
        (pydcop/algorithms/maxsum_dynamic.py:80-104) its table replaces the old one on the device
        and the iteration carries on; over other variables (DynamicFactorComputation,
        maxsum_dynamic.py:234-271) the factor's edges are re-laid out and the messages of the
        surviving edges carried over (pydcop_amd/dynamic.py)."""
        factor = getattr(self, "_current_factor", None) or self.computation_def.node.factor
        self._session.update_factor(self.name, factor, fn)
        self._current_factor = fn


class MaxSumGpuVariableComputation(_ProxyMixin, VariableComputation):
    """Stands for a MaxSumVariableComputation (pydcop/algorithms/maxsum.py:450):
    publishes the value the GPU sweep selected for its variable."""

    def __init__(self, comp_def: ComputationDef):
        super().__init__(comp_def.node.variable, comp_def)
        self._init_proxy(comp_def)

    def _publish(self):
        val, cost = self._session.value_of(self.name)
        self.value_selection(val, cost)


def build_computation(comp_def: ComputationDef):
    if comp_def.node.type == "VariableComputation":
        return MaxSumGpuVariableComputation(comp_def)
    if comp_def.node.type == "FactorComputation":
        return MaxSumGpuFactorComputation(comp_def)
    raise ValueError("maxsum_gpu: unsupported computation node type " + str(comp_def.node.type))
