"""The two halves meet on hardware: the UNMODIFIED reference (`pydcop solve`, the real
orchestrator, agents and YAML loader -- pydcop/commands/solve.py:444-632,
infrastructure/orchestratedagents.py:265-290, agents.py:785-838) with `--algo maxsum_gpu` on the
REAL libmaxsum_hip.so.

The reference reaches the GPU box as the git-ignored archive `oracle/_ref/` that
`__graft_entry__.build()` packs in the build container (`oracle/stage_reference.py`); it is used
here only as the CALLER of the plug-in.  Expected results are the reference's own:
tests/dcop_cli/test_solve.py:39-72 (secp_simple1 -> l1=0, l2=3, l3=4, m1=3), :100-130
(graph_coloring1 -> v1=R, v2=G, v3=R), SURVEY.md section 8c (costs -0.1 / 2.3)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.stage_reference import locate

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(locate() is None,
                                 reason="no reference: oracle/_ref/pydcop_reference.tar.gz was not staged by build()")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = locate()

# the child reports every native library of ours it mapped, so that the test can tell the HIP
# build ran (and no emulated engine, no oracle)
CHILD = (
    "import atexit, sys\n"
    "def _report():\n"
    "    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if '.so' in l and\n"
    "                   ('maxsum' in l or 'oracle' in l or 'emu' in l)})\n"
    "    sys.stderr.write('LOADED ' + repr(libs) + '\\n')\n"
    "atexit.register(_report)\n"
    "from pydcop_amd import plugin\n"
    "plugin.main()\n")


def run_cli(args, env_extra=None, timeout=600):
    assert REF, "no reference: oracle/_ref/pydcop_reference.tar.gz was not staged by build()"
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, REF, env.get("PYTHONPATH", "")])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, "-c", CHILD] + list(args), capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    loaded = [l for l in out.stderr.splitlines() if l.startswith("LOADED ")]
    assert loaded, out.stderr[-2000:]
    libs = eval(loaded[-1][7:])
    assert any(l.endswith("pydcop_amd/csrc/libmaxsum_hip.so") for l in libs), libs
    assert not any("emu" in l or "oracle" in l for l in libs), libs
    s = out.stdout
    return json.loads(s[s.index("{"):])


@pytest.mark.parametrize("instance,expected,cost", [
    ("graph_coloring1.yaml", {"v1": "R", "v2": "G", "v3": "R"}, -0.1),
    ("secp_simple1.yaml", {"l1": 0, "l2": 3, "l3": 4, "m1": 3}, 2.3),
    ("graph_coloring_tuto.yaml", None, 12),
])
@pytest.mark.parametrize("dist", ["adhoc", "oneagent"])
def test_pydcop_solve_cli_on_the_hip_library(instance, expected, cost, dist):
    if dist == "oneagent" and instance != "graph_coloring1.yaml":
        pytest.skip("oneagent needs one agent per computation (oneagent.py:127-130)")
    res = run_cli(["-t", "30", "solve", "--algo", "maxsum_gpu", "-p", "stop_cycle:30", "-p", "noise:0",
                   "-d", dist, os.path.join(REF, "tests", "instances", instance)])
    if expected is not None:
        assert res["assignment"] == expected
    assert res["status"] == "FINISHED" and res["violation"] == 0 and res["cycle"] == 30
    assert abs(res["cost"] - cost) < 1e-9
    for key in ("agt_metrics", "msg_count", "msg_size", "time"):   # orchestrator.py:1262-1272
        assert key in res


def test_reference_maxsum_and_maxsum_gpu_agree_through_the_same_cli():
    """Same YAML, same CLI, `--algo maxsum` (the reference's thread agents, ended by the timeout)
    and `--algo maxsum_gpu`: same assignment and cost (tests/dcop_cli/test_solve.py:100-130)."""
    inst = os.path.join(REF, "tests", "instances", "graph_coloring1.yaml")
    gpu = run_cli(["-t", "30", "solve", "--algo", "maxsum_gpu", "-p", "stop_cycle:30", "-p", "noise:0", "-d", "adhoc", inst])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, REF]), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-m", "pydcop_amd.plugin", "-t", "3", "solve", "--algo", "maxsum",
                          "-p", "noise:0", "-d", "adhoc", inst], capture_output=True, text=True, timeout=120,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    ref = json.loads(out.stdout[out.stdout.index("{"):])
    assert ref["status"] == "TIMEOUT"
    assert gpu["assignment"] == ref["assignment"] == {"v1": "R", "v2": "G", "v3": "R"}
    assert abs(gpu["cost"] - ref["cost"]) < 1e-9


def test_generated_2k_variable_yaml_with_the_fast_graph_builder(tmp_path, oracle_built):
    """A generated 2 000-variable colouring written as YAML, loaded by the reference's own loader,
    graph built by GRAPH_TYPE factor_graph_fast (MAXSUM_GPU_GRAPH=fast), all 6 000 computations on
    one agent through a distribution file; the selection must be what the oracle selects on the
    generator's arrays after the same 25 cycles, the cost what DCOP.solution_cost makes of it."""
    from yaml_instances import write_coloring_yaml
    from pydcop_amd import generators as G
    from pydcop_amd.graph import Params
    g = G.random_coloring(2000, seed=5, names=True)
    dcop, dist = str(tmp_path / "gen.yaml"), str(tmp_path / "gen_dist.yaml")
    write_coloring_yaml(g, dcop, dist)
    res = run_cli(["-t", "300", "solve", "--algo", "maxsum_gpu", "-p", "stop_cycle:25", "-p", "noise:0",
                   "-d", dist, dcop], env_extra={"MAXSUM_GPU_GRAPH": "fast"})
    assert res["status"] == "FINISHED" and res["cycle"] == 25
    ora = oracle_built.OracleMaxSum(g, Params())
    ora.run(25)
    idx, _ = ora.assignment()
    got = np.array([res["assignment"][n] for n in g.var_names])
    assert np.array_equal(got, idx)
    cost, violations = ora.eval_cost()
    assert res["violation"] == violations == 0
    assert abs(res["cost"] - cost) <= 1e-9 * max(1.0, abs(cost))
    ora.close()


def test_reference_generated_meeting_scheduling_through_the_cli(tmp_path, oracle_built):
    """The reference's OWN generator and CLI end to end: the code of `pydcop generate meetings` (the PEAV model,
    pydcop/commands/generators/meetingscheduling.py:211-365: binary utility / conflict and equality tables over
    `slots - length + 2` values, unary tables for single-event resources) writes the YAML, `pydcop solve --algo
    maxsum_gpu` runs it on the MI355X -- every factor of it on the lane-grid kernel (bin_box.h) -- and the assignment
    and cost are what the oracle computes on the arrays `pydcop_amd.api.compile_dcop` makes of the same file."""
    # (the generator's own code, in this process: its `NAryMatrixRelation.set_value_for_assignment` needs `ndarray.itemset`,
    # gone in numpy 2 -- oracle/ref_harness.install_shims carries the stand-in; the YAML is written by the reference's dcop_yaml)
    import random
    from oracle import ref_harness
    ref_harness.install_shims()
    from pydcop.commands.generators import meetingscheduling as M
    from pydcop.dcop.dcop import DCOP
    from pydcop.dcop.objects import AgentDef
    from pydcop.dcop.yamldcop import dcop_yaml
    random.seed(11)
    slots, events, resources = M.generate_problem_definition(9, 5, 10, 8, 4, 3)   # slots, resources, max value, events, max length, max resources
    variables, constraints, agents = M.peav_model(slots, events, resources, 10 * 9 * 5)   # penalty as generate() sets it (:222)
    dcop = DCOP("MeetingSceduling", objective="max",
                domains={v.domain.name: v.domain for v in variables.values()},
                variables={v.name: v for v in variables.values()}, constraints=constraints,
                agents={a: AgentDef(a, capacity=100000) for a in agents})
    dcop_path = str(tmp_path / "meetings.yaml")
    with open(dcop_path, "w", encoding="utf-8") as f:
        f.write(dcop_yaml(dcop))
    res = run_cli(["-t", "120", "solve", "--algo", "maxsum_gpu", "-p", "stop_cycle:20", "-p", "noise:0", "-d", "adhoc", dcop_path])
    assert res["status"] == "FINISHED" and res["cycle"] == 20
    # the same file, loaded by the reference's loader in this process, compiled to the flat arrays, on the oracle
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop_amd.api import compile_dcop
    from pydcop_amd.compile import assignment_to_values
    from pydcop_amd.engine import MaxSumEngine
    from pydcop_amd.graph import Params
    dcop = load_dcop_from_file([dcop_path])
    assert dcop.objective == "max"
    g = compile_dcop(dcop, noise=0.0)
    assert int(g.dom_size.max()) > 4                        # beyond the register classes
    with MaxSumEngine(g, Params(mode="max")) as eng:
        k = eng.factor_kernels()
        assert k["lane_grid"] == g.n_factors and k["generic"] == 0, k
    ora = oracle_built.OracleMaxSum(g, Params(mode="max"))
    ora.run(20)
    idx, _ = ora.assignment()
    assert assignment_to_values(g, idx) == res["assignment"]
    cost, violations = ora.eval_cost()
    assert res["violation"] == violations
    assert abs(res["cost"] - cost) <= 1e-9 * max(1.0, abs(cost))
    ora.close()
