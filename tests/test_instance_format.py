"""the compact binary instance format (.npz) beside YAML --
`FlatGraph.save / load`, `pydcop_amd.api.solve_flat` and the CLI on an .npz instance
(no pyDCOP import on that path).  Emulated engine: no GPU here."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pydcop_amd import generators as G
from pydcop_amd.graph import FlatGraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def emu_lib():
    from emu.build_emu import build
    return build()


def test_save_load_round_trip(tmp_path):
    g = G.random_mixed(60, 90, seed=4)
    g.init_idx = np.where(np.arange(g.n_vars) % 7 == 0, 0, -1).astype(np.int32)
    path = str(tmp_path / "inst.npz")
    g.save(path, objective="max", note="round trip")
    h, header = FlatGraph.load(path)
    assert header["objective"] == "max" and header["meta"] == {"note": "round trip"}
    for k in FlatGraph._ARRAYS + ("init_idx",):
        a, b = getattr(g, k), getattr(h, k)
        assert a.dtype == b.dtype and np.array_equal(a, b), k
    assert h.var_names == g.var_names and h.factor_names == g.factor_names
    assert [list(d) for d in h.domains] == [list(d) for d in g.domains]
    assert h.var_owned is None and h.factor_owned is None


def test_load_rejects_foreign_files(tmp_path):
    p = str(tmp_path / "x.npz")
    np.savez(p, a=np.arange(3))
    with pytest.raises(ValueError, match="not a maxsum_gpu instance"):
        FlatGraph.load(p)
    g = G.random_coloring(10, seed=0)
    g.save(p)
    z = dict(np.load(p))
    z["edge_var"] = z["edge_var"][:-1]
    np.savez(p, **z)
    with pytest.raises(ValueError):
        FlatGraph.load(p)


def test_solve_flat_and_cli_on_npz(emu_lib, tmp_path):
    from pydcop_amd.api import solve_flat
    from pydcop_amd.engine import MaxSumEngine
    from pydcop_amd.graph import Params
    g = G.random_coloring(300, seed=2)
    path = str(tmp_path / "col.npz")
    g.save(path, objective="min")
    res = solve_flat(FlatGraph.load(path)[0], "min", 25, lib_path=emu_lib, cost_every=10)
    with MaxSumEngine(g, Params(), lib_path=emu_lib) as e:
        e.run(25)
        idx = e.assignment()[0]
        cost, viol = e.eval_cost(infinity=10000)
    assert res["cost"] == cost and res["violation"] == viol and res["cycle"] == 25
    assert [c[0] for c in res["cost_curve"]] == [10, 20, 25]
    assert res["assignment"] == {n: g.domains[i][int(idx[i])] for i, n in enumerate(g.var_names)}
    # the CLI, in a process that never imports pyDCOP
    code = ("import sys, runpy; sys.argv = ['api', '-c', '25', '-p', 'precision:f64', %r]; "
            "from pydcop_amd import engine; engine.register_test_engine(%r, make_default=True); "
            "runpy.run_module('pydcop_amd.api', run_name='__main__'); "
            "assert 'pydcop' not in sys.modules" % (path, emu_lib))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout)
    assert out["status"] == "FINISHED" and out["cycle"] == 25 and out["violation"] == viol
    assert abs(out["cost"] - g_cost(g, idx)) < 1e-9 * max(1.0, abs(cost))


def g_cost(g, idx):
    """numpy evaluation of the assignment: sum of table entries + variable costs."""
    cost = float(g.var_cost[g.cost_off[:-1] + idx].sum())
    for f in range(g.n_factors):
        e0, e1 = g.factor_rowptr[f], g.factor_rowptr[f + 1]
        lin = 0
        for e in range(e0, e1):
            lin = lin * g.dom_size[g.edge_var[e]] + idx[g.edge_var[e]]
        cost += float(g.tables[g.table_off[f] + lin])
    return cost


@pytest.mark.parametrize("algo", ["amaxsum", "dsa", "mgm"])
def test_solve_flat_other_algorithms(algo, emu_lib, tmp_path):
    """`solve_flat(algo=...)` and `python -m pydcop_amd.api -a ...` = the engine of that algorithm used
    directly (amaxsum: generations under FIFO delivery; dsa: variant / probability / seed passed on)."""
    from pydcop_amd.amaxsum import AMaxSumEngine
    from pydcop_amd.api import solve_flat
    from pydcop_amd.dsa import DsaEngine
    from pydcop_amd.graph import Params
    from pydcop_amd.mgm import MgmEngine
    g = G.random_coloring(120, seed=4)
    kw = dict(variant="C", probability=0.6, seed=9) if algo == "dsa" else {}
    res = solve_flat(g, "min", 9, lib_path=emu_lib, cost_every=4, algo=algo, start_messages="leafs_vars", **kw)
    make = {"amaxsum": lambda: AMaxSumEngine(g, Params(start_messages="leafs_vars"), lib_path=emu_lib),
            "dsa": lambda: DsaEngine(g, Params(), lib_path=emu_lib, **kw),
            "mgm": lambda: MgmEngine(g, Params(), lib_path=emu_lib)}[algo]
    with make() as e:
        e.run(9)
        idx = e.assignment()[0]
        cost, viol = e.eval_cost(infinity=10000)
    assert res["cost"] == cost and res["violation"] == viol
    assert [c[0] for c in res["cost_curve"]] == [4, 8, 9]
    assert res["assignment"] == {n: g.domains[i][int(idx[i])] for i, n in enumerate(g.var_names)}
    with pytest.raises(ValueError):
        solve_flat(g, "min", 3, lib_path=emu_lib, algo=algo, devices=2)
    path = str(tmp_path / "col.npz")
    g.save(path, objective="min")
    extra = ["-p", "variant:C", "-p", "probability:0.6", "-p", "seed:9"] if algo == "dsa" else []
    code = ("import sys, runpy; sys.argv = ['api', '-c', '9', '-a', %r, '-p', 'start_messages:leafs_vars'] + %r + [%r]; "
            "from pydcop_amd import engine; engine.register_test_engine(%r, make_default=True); "
            "runpy.run_module('pydcop_amd.api', run_name='__main__')" % (algo, extra, path, emu_lib))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout)
    assert out["violation"] == viol and abs(out["cost"] - cost) < 1e-9 * max(1.0, abs(cost))


def test_solve_flat_rejects_unknown_algorithm(emu_lib):
    from pydcop_amd.api import solve_flat
    with pytest.raises(ValueError, match="algo must be one of"):
        solve_flat(G.random_coloring(10, seed=0), "min", 1, lib_path=emu_lib, algo="dpop")


def test_value_rank_of_the_domains():
    """FlatGraph.value_rank: what DSA / MGM break the start-up cost ties of a variable without
    neighbours on (the reference's optimal_cost_value compares (cost, value) tuples,
    relations.py:1661-1665) -- None whenever the index order already is the value order."""
    g = G.random_coloring(6, seed=0)
    assert g.value_rank() is None                      # range(D): ascending as written
    g.domains = None
    assert g.value_rank() is None                      # no values known
    g = G.random_coloring(6, seed=0)
    g.domains = [["R", "G", "B"], ["B", "G", "R"], [2, 0, 1], [0, 1, 2], ["a", "b", "c"], [3.5, -1.0, 2.0]]
    np.testing.assert_array_equal(g.value_rank().reshape(6, 3),
                                  [[2, 1, 0], [0, 1, 2], [2, 0, 1], [0, 1, 2], [0, 1, 2], [2, 0, 1]])
    g.domains[4] = ["a", 1, None]                      # values that do not compare: the reference would raise on a tie
    assert g.value_rank() is None
