"""The drop-in boundary: `maxsum_gpu` behind an UNMODIFIED pyDCOP (SURVEY.md
section 8b).  Needs the reference checkout (build container only); the engine is
the emulated build, because there is no GPU here -- the plumbing under test
(plugin discovery, parameter coercion, ComputationDef -> flat arrays, proxies,
value_selection/finished on the agents' threads, orchestrator result JSON) is
the same with the HIP library.

Mirrors the reference's own tests for this path: tests/dcop_cli/test_solve.py:39-130
(graph_coloring1 -> v1=R, v2=G, v3=R; secp_simple1 -> l1=0, l2=3, l3=4, m1=3) and
tests/api/test_api_solve.py:37-45.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle.stage_reference import locate as _locate_reference  # noqa: E402
REF = _locate_reference() or "/root/reference"
INST = os.path.join(REF, "tests", "instances")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pydcop")),
                                reason="the pyDCOP reference checkout is not on this machine")


@pytest.fixture(scope="module")
def emu_lib():
    from emu.build_emu import build
    return build()


@pytest.fixture(scope="module")
def pydcop_ready(emu_lib):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from pydcop_amd import plugin
    plugin.install()
    from pydcop.algorithms import load_algorithm_module
    mod = load_algorithm_module("maxsum_gpu")
    from pydcop_amd import engine
    before = engine.DEFAULT_LIB
    engine.register_test_engine(emu_lib, make_default=True)  # emulated engine (no GPU in this container)
    yield mod
    engine.DEFAULT_LIB = before


def test_discovered_like_a_builtin_algorithm(pydcop_ready):
    from pydcop.algorithms import list_available_algorithms, load_algorithm_module
    assert "maxsum_gpu" in list_available_algorithms()
    ref = load_algorithm_module("maxsum")
    mod = pydcop_ready
    assert mod.GRAPH_TYPE == ref.GRAPH_TYPE == "factor_graph"
    ref_params = {p.name: (p.type, p.values, p.default_value) for p in ref.algo_params}
    mine = {p.name: (p.type, p.values, p.default_value) for p in mod.algo_params}
    for name, spec in ref_params.items():  # same parameters, same defaults
        assert mine[name] == spec


def test_amaxsum_gpu_is_discovered_with_the_reference_parameters(pydcop_ready):
    from pydcop.algorithms import list_available_algorithms, load_algorithm_module
    assert "amaxsum_gpu" in list_available_algorithms()
    ref, mod = load_algorithm_module("amaxsum"), load_algorithm_module("amaxsum_gpu")
    assert mod.GRAPH_TYPE == ref.GRAPH_TYPE
    refp = {p.name: (p.type, p.values, p.default_value) for p in ref.algo_params}
    mine = {p.name: (p.type, p.values, p.default_value) for p in mod.algo_params}
    assert all(mine[k] == v for k, v in refp.items())


def test_param_validation_matches_reference(pydcop_ready):
    from pydcop.algorithms import AlgorithmDef
    a = AlgorithmDef.build_with_default_param("maxsum_gpu", {"stop_cycle": "30", "damping": "0.7"}, mode="min")
    assert a.params["stop_cycle"] == 30 and a.params["damping"] == 0.7
    with pytest.raises(ValueError):
        AlgorithmDef.build_with_default_param("maxsum_gpu", {"damping_nodes": "sometimes"})
    with pytest.raises(ValueError):
        AlgorithmDef.build_with_default_param("maxsum_gpu", {"no_such_param": 1})


def test_footprint_and_load_formulas_equal_reference(pydcop_ready):
    from pydcop.algorithms import load_algorithm_module
    from pydcop.computations_graph import factor_graph
    from pydcop.dcop.yamldcop import load_dcop_from_file
    ref = load_algorithm_module("maxsum")
    dcop = load_dcop_from_file([os.path.join(INST, "graph_coloring_tuto.yaml")])
    cg = factor_graph.build_computation_graph(dcop)
    for node in cg.nodes:
        assert pydcop_ready.computation_memory(node) == ref.computation_memory(node)
        for n in node.neighbors:
            assert pydcop_ready.communication_load(node, n) == ref.communication_load(node, n)


@pytest.mark.parametrize("instance,expected", [
    ("graph_coloring1.yaml", {"v1": "R", "v2": "G", "v3": "R"}),
    ("secp_simple1.yaml", {"l1": 0, "l2": 3, "l3": 4, "m1": 3}),
])
def test_api_solve(pydcop_ready, instance, expected):
    """pydcop.infrastructure.run.solve with thread agents (reference:
    tests/api/test_api_solve.py, tests/dcop_cli/test_solve.py)."""
    from pydcop.algorithms import AlgorithmDef
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop.infrastructure.run import solve
    dcop = load_dcop_from_file([os.path.join(INST, instance)])
    algo = AlgorithmDef.build_with_default_param(
        "maxsum_gpu", {"stop_cycle": 20, "noise": 0}, mode=dcop.objective)
    assignment = solve(dcop, algo, "adhoc", timeout=5)
    assert assignment == expected


@pytest.mark.parametrize("instance,expected", [
    ("graph_coloring1.yaml", {"v1": "R", "v2": "G", "v3": "R"}),
    ("secp_simple1.yaml", {"l1": 0, "l2": 3, "l3": 4, "m1": 3}),
])
def test_api_solve_on_two_devices(pydcop_ready, instance, expected, tmp_path, monkeypatch):
    """`-p devices:2`: the plugin partitions the compiled graph over two (emulated) GPUs --
    LocalShardedMaxSum, the engines' own exchange over the fake RCCL -- same answers."""
    from emu.build_emu import build_fake_rccl
    from pydcop.algorithms import AlgorithmDef
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop.infrastructure.run import solve
    monkeypatch.setenv("EMU_HIP_DEVICES", "2")
    monkeypatch.setenv("MAXSUM_RCCL_LIB", build_fake_rccl())
    monkeypatch.setenv("FAKE_RCCL_DIR", str(tmp_path))
    dcop = load_dcop_from_file([os.path.join(INST, instance)])
    algo = AlgorithmDef.build_with_default_param(
        "maxsum_gpu", {"stop_cycle": 20, "noise": 0, "devices": 2}, mode=dcop.objective)
    assert solve(dcop, algo, "adhoc", timeout=5) == expected


@pytest.mark.parametrize("instance", ["graph_coloring1.yaml", "graph_coloring_tuto.yaml", "secp_simple1.yaml",
                                      "graph_coloring_3agts_10vars.yaml"])
def test_mgm_gpu_equals_the_reference_mgm(pydcop_ready, instance):
    """`--algo mgm_gpu` through the unmodified orchestrator / agents == the reference's own
    MgmComputation objects with the same deterministic choices (first value at start, first of
    equally good values), after the same number of rounds; module attributes like the reference's."""
    from oracle.ref_harness import run_reference_mgm
    from pydcop.algorithms import AlgorithmDef, load_algorithm_module
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop.infrastructure.run import solve
    ref, mod = load_algorithm_module("mgm"), load_algorithm_module("mgm_gpu")
    assert mod.GRAPH_TYPE == ref.GRAPH_TYPE == "constraints_hypergraph"
    refp = {p.name: (p.type, p.values, p.default_value) for p in ref.algo_params}
    mine = {p.name: (p.type, p.values, p.default_value) for p in mod.algo_params}
    assert all(mine[k] == v for k, v in refp.items())
    dcop = load_dcop_from_file([os.path.join(INST, instance)])
    algo = AlgorithmDef.build_with_default_param("mgm_gpu", {"stop_cycle": 9}, mode=dcop.objective)
    got = solve(dcop, algo, "adhoc", timeout=5)
    dcop2 = load_dcop_from_file([os.path.join(INST, instance)])
    want, _, _ = run_reference_mgm(dcop2, 8)
    assert got == want


@pytest.mark.parametrize("instance,variant", [("graph_coloring1.yaml", "B"), ("graph_coloring_tuto.yaml", "A"),
                                              ("graph_coloring_3agts_10vars.yaml", "C")])
def test_dsa_gpu_equals_the_reference_dsa(pydcop_ready, instance, variant):
    """`--algo dsa_gpu` through the unmodified orchestrator / agents == the reference's own
    DsaComputation objects drawing from the same keyed generator (same seed), after the same number of
    cycles; module attributes like the reference's."""
    from oracle.ref_harness import run_reference_dsa
    from pydcop.algorithms import AlgorithmDef, load_algorithm_module
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop.infrastructure.run import solve
    ref, mod = load_algorithm_module("dsa"), load_algorithm_module("dsa_gpu")
    assert mod.GRAPH_TYPE == ref.GRAPH_TYPE == "constraints_hypergraph"
    assert (mod.UNIT_SIZE, mod.HEADER_SIZE) == (ref.UNIT_SIZE, ref.HEADER_SIZE)
    refp = {p.name: (p.type, p.values, p.default_value) for p in ref.algo_params}
    mine = {p.name: (p.type, p.values, p.default_value) for p in mod.algo_params}
    assert all(mine[k] == v for k, v in refp.items())
    dcop = load_dcop_from_file([os.path.join(INST, instance)])
    algo = AlgorithmDef.build_with_default_param("dsa_gpu", {"stop_cycle": 12, "variant": variant, "seed": 4},
                                                 mode=dcop.objective)
    got = solve(dcop, algo, "adhoc", timeout=5)
    dcop2 = load_dcop_from_file([os.path.join(INST, instance)])
    want, _, _ = run_reference_dsa(dcop2, 12, variant=variant, seed=4)
    assert got == want


def test_cli_solve_json(emu_lib, tmp_path):
    """`pydcop solve --algo maxsum_gpu` through the launcher, result JSON of the
    unmodified orchestrator (docs/tutorials/analysing_results.rst:31-48)."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from pydcop_amd import plugin, engine; plugin.install()\n"
        "engine.register_test_engine(%r, make_default=True)\n"
        "sys.argv = ['pydcop', '-t', '20', 'solve', '--algo', 'maxsum_gpu', '-p', 'stop_cycle:20',\n"
        "            '-p', 'noise:0', '-d', 'adhoc', %r]\n"
        "from pydcop import dcop_cli; dcop_cli.main()\n"
    ) % (ROOT, REF, emu_lib, os.path.join(INST, "graph_coloring1.yaml"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    start = out.stdout.index("{")
    res = json.loads(out.stdout[start:])
    assert res["assignment"] == {"v1": "R", "v2": "G", "v3": "R"}
    assert res["status"] == "FINISHED"
    assert res["violation"] == 0 and abs(res["cost"] - (-0.1)) < 1e-9
    assert res["cycle"] == 20


UNSORTED_YAML = """name: unsorted domains
objective: {objective}

domains:
  colors:
    values: [R, G, B]
    type: 'color'

variables:
  v1:
    domain: colors
    cost_function: 0.25 if v1 == 'G' else 0.5
  v2:
    domain: colors
    cost_function: 0.5 if v2 == 'R' else 0.25
  v3:
    domain: colors
    cost_function: 0.125 if v3 == 'R' else 0.125
  lonely1:
    domain: colors
    cost_function: 0.75 if lonely1 == 'G' else 0.5
  lonely2:
    domain: colors
    cost_function: 0.5 if lonely2 == 'R' else 0.5
  lonely3:
    domain: colors
    cost_function: 0.25 if lonely3 == 'B' else 0.75

constraints:
  diff_1_2:
    type: intention
    function: 1 if v1 == v2 else 0
  diff_2_3:
    type: intention
    function: 1 if v3 == v2 else 0

agents:
  a1:
    capacity: 1000
  a2:
    capacity: 1000
"""


@pytest.mark.parametrize("objective", ["min", "max"])
@pytest.mark.parametrize("algo", ["mgm", "dsa"])
def test_isolated_variables_break_cost_ties_on_the_value_like_the_reference(pydcop_ready, tmp_path, objective, algo):
    """Domains written in a non-ascending order (R, G, B) and variables WITHOUT neighbours whose own costs
    tie: the reference's optimal_cost_value takes min / max over (cost, value) tuples
    (relations.py:1661-1665) -- `lonely1` ties R and B, `lonely2` all three.  The plug-ins get the order
    of the values from the DCOP's domains (compile.py -> FlatGraph.value_rank -> mxs_*_set_value_rank)."""
    from oracle.ref_harness import run_reference_dsa, run_reference_mgm
    from pydcop.algorithms import AlgorithmDef
    from pydcop.dcop.yamldcop import load_dcop_from_file
    from pydcop.infrastructure.run import solve
    path = tmp_path / "unsorted.yaml"
    path.write_text(UNSORTED_YAML.format(objective=objective))
    dcop = load_dcop_from_file([str(path)])
    params = {"stop_cycle": 6} if algo == "mgm" else {"stop_cycle": 6, "variant": "A", "seed": 3}
    got = solve(dcop, AlgorithmDef.build_with_default_param(algo + "_gpu", params, mode=dcop.objective), "adhoc", timeout=5)
    dcop2 = load_dcop_from_file([str(path)])
    if algo == "mgm":
        want, _, _ = run_reference_mgm(dcop2, 5)
    else:
        want, _, _ = run_reference_dsa(dcop2, 6, variant="A", seed=3)
    assert got == want
    # smallest / largest VALUE among the tied ones (lonely1: R and B tie at the minimum, G alone is the maximum)
    assert got["lonely2"] == {"min": "B", "max": "R"}[objective]
    assert got["lonely1"] == {"min": "B", "max": "G"}[objective]
