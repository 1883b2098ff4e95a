"""TEST stand-in for what an agent does to a computation (the real classes:
pydcop/infrastructure/computations.py:229 ComputationException, :277-569
MessagePassingComputation, :840-1000 DcopComputation, :1003-1100 VariableComputation; the real
caller: pydcop/infrastructure/agents.py:785-838).  Single-threaded: `MiniAgent.pump()` plays
the agent's loop -- it fires the periodic actions that are due."""
import time


class ComputationException(Exception):
    pass


class MessagePassingComputation:
    def __init__(self, name):
        self._name = name
        self.is_running = False
        self.is_finished = False
        self._periodic = {}   # handle -> [period, callback, next time]
        self._handles = 0
        self._cycles = 0
        self.cycle_events = []

    @property
    def name(self):
        return self._name

    def start(self):
        self.is_running = True
        self.on_start()

    def stop(self):
        if self.is_running:
            self.is_running = False
            self.on_stop()

    def on_start(self):
        pass

    def on_stop(self):
        pass

    def finished(self):
        self.is_finished = True

    def add_periodic_action(self, period, cb):
        self._handles += 1
        self._periodic[self._handles] = [period, cb, time.monotonic() + period]
        return self._handles

    def remove_periodic_action(self, handle):
        self._periodic.pop(handle, None)

    def run_due_actions(self):
        now = time.monotonic()
        for h, rec in list(self._periodic.items()):
            if h in self._periodic and now >= rec[2]:
                rec[2] = now + rec[0]
                rec[1]()


class DcopComputation(MessagePassingComputation):
    def __init__(self, name, comp_def):
        super().__init__(name)
        self.computation_def = comp_def

    @property
    def cycle_count(self):
        return self._cycles

    def new_cycle(self):
        self._cycles += 1
        self.cycle_events.append(self.cycle_count)

    def footprint(self):
        return 0.0


class VariableComputation(DcopComputation):
    def __init__(self, variable, comp_def):
        super().__init__(variable.name, comp_def)
        self._variable = variable
        self.current_value = None
        self.current_cost = None
        self.selections = []

    @property
    def variable(self):
        return self._variable

    def value_selection(self, val, cost=0):
        if val != self.current_value:
            self.selections.append((val, cost, self.cycle_count))
        self.current_value, self.current_cost = val, cost


class MiniAgent:
    """Hosts computations the way OrchestratedAgent does in thread mode, on the calling thread."""

    def __init__(self, computations):
        self.computations = list(computations)

    def start_all(self):
        for c in self.computations:
            c.start()

    def pump(self, until, timeout=60.0, sleep=0.005):
        t0 = time.monotonic()
        while not until():
            if time.monotonic() - t0 > timeout:
                raise TimeoutError("computations did not finish")
            for c in self.computations:
                if c.is_running:
                    c.run_due_actions()
            time.sleep(sleep)

    def stop_all(self):
        for c in self.computations:
            c.stop()
