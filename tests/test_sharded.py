"""The multi-GPU path: partition, shard construction, halo pack/unpack and the
per-cycle exchange -- against a single engine sweeping the whole graph, which it
must reproduce bit for bit (shards inherit the global factor and link orders).

CPU: emulated engines, single-process shards + a 2-rank gloo run.
GPU: k engines on one MI355X, and the nccl (= RCCL) code path with world_size 1.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from parity_common import parity_cases
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params
from pydcop_amd.partition import build_shard, cut_statistics, partition_variables
from shard_harness import LocalShards, make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def emu_lib():
    from emu.build_emu import build
    return build()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_is_balanced_and_local():
    g, _ = make_case("ising")
    part = partition_variables(g, 4)
    st = cut_statistics(g, part)
    assert set(np.unique(part)) == {0, 1, 2, 3}
    assert st["edge_imbalance"] < 1.15
    # a 16x12 torus cut in 4 has far fewer cut factors than a random split
    rnd = cut_statistics(g, np.random.default_rng(0).integers(0, 4, g.n_vars).astype(np.int32))
    assert st["cut_factors"] < 0.5 * rnd["cut_factors"]
    g2, _ = make_case("coloring")
    for k in (2, 3, 8):
        p = partition_variables(g2, k)
        assert p.min() == 0 and p.max() == k - 1
        assert cut_statistics(g2, p)["edge_imbalance"] < 1.2


def test_shards_cover_the_graph():
    g, _ = make_case("mixed_max")
    k = 3
    part = partition_variables(g, k)
    shards = [build_shard(g, part, r, k) for r in range(k)]
    owned = np.concatenate([s.local_vars[:s.n_owned] for s in shards])
    assert np.array_equal(np.sort(owned), np.arange(g.n_vars))
    counted = np.concatenate([s.local_factors[s.graph.factor_owned == 1] for s in shards])
    assert np.array_equal(np.sort(counted), np.arange(g.n_factors))
    for r, s in enumerate(shards):
        s.graph.validate()
        for q in range(k):  # what r sends to q is what q expects from r, element for element
            assert s.send_counts[q] == shards[q].recv_counts[r]
        assert s.send_counts[r] == 0 and s.recv_counts[r] == 0


def _check_against_single(g, kw, k, lib_path, device="cpu", steps=(0, 1, 2, 7, 20), dtype="f64"):
    p = Params(dtype=dtype, **kw)
    one = MaxSumEngine(g, p, lib_path=lib_path)
    many = LocalShards(g, p, k, lib_path=lib_path, device=device)
    for n in steps:
        one.run(n)
        many.run(n)
        i1, b1 = one.assignment()
        i2, b2 = many.assignment()
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(b1, b2)
        c1, c2 = one.eval_cost(), many.eval_cost()
        assert c1[1] == c2[1] and abs(c1[0] - c2[0]) <= 1e-9 * max(1.0, abs(c1[0]))
    one.close()
    many.close()


@pytest.mark.parametrize("case", ["coloring", "mixed_max", "ising", "coloring_deg9"])
@pytest.mark.parametrize("k", [2, 5])
def test_local_shards_equal_single_engine_emu(case, k, emu_lib):
    g, kw = make_case(case)
    _check_against_single(g, kw, k, emu_lib)


def test_local_shards_parity_cases_emu(emu_lib):
    for name, make, kw in parity_cases()[:4] + parity_cases()[9:11]:
        _check_against_single(make(), kw, 3, emu_lib, steps=(1, 6))


def _run_ranks(world, lib, case, steps, tmp_path, timeout=600):
    port = _free_port()
    out = str(tmp_path / "sharded.npz")
    procs = []
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for r in range(world):
        cmd = [sys.executable, os.path.join(ROOT, "tests", "shard_harness.py"), str(world), str(r),
               str(port), lib or "-", out, case] + [str(s) for s in steps]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for p2 in procs:
                p2.kill()
            raise
        logs.append(o.decode()[-2000:])
    assert all(pr.returncode == 0 for pr in procs), "\n".join(logs)
    return np.load(out)


@pytest.mark.parametrize("case", ["coloring", "mixed_max"])
def test_gloo_two_ranks_equal_single_engine(case, emu_lib, tmp_path):
    """pydcop_amd.sharded.ShardedMaxSum over torch.distributed (gloo, world 2)."""
    steps = [1, 5, 14]
    z = _run_ranks(2, emu_lib, case, steps, tmp_path)
    g, kw = make_case(case)
    one = MaxSumEngine(g, Params(**kw), lib_path=emu_lib)
    done = 0
    for n in steps:
        one.run(n)
        done += n
        i1, b1 = one.assignment()
        np.testing.assert_array_equal(z[f"idx_{done}"], i1)
        np.testing.assert_array_equal(z[f"bel_{done}"], b1)
        c, v = one.eval_cost()
        assert z[f"cost_{done}"][1] == v and abs(z[f"cost_{done}"][0] - c) <= 1e-9 * max(1, abs(c))


# ---- on a real MI355X -----------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 4])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_local_shards_equal_single_engine_gpu(k, dtype):
    for case in ("coloring", "mixed_max", "ising", "coloring_deg9"):
        g, kw = make_case(case)
        _check_against_single(g, kw, k, None, device="cuda", dtype=dtype)
    g, kw = make_case("coloring_50k")
    _check_against_single(g, kw, k, None, device="cuda", steps=(30,), dtype=dtype)


@pytest.mark.gpu
def test_nccl_world1_code_path(tmp_path):
    """ShardedMaxSum with backend nccl (RCCL): external stream + torch-owned halo
    tensors bound to the engine; world_size 1 (one GPU on the test box)."""
    z = _run_ranks(1, None, "coloring_50k", [25], tmp_path)
    g, kw = make_case("coloring_50k")
    one = MaxSumEngine(g, Params(**kw))
    one.run(25)
    np.testing.assert_array_equal(z["idx_25"], one.assignment()[0])
    np.testing.assert_array_equal(z["bel_25"], one.assignment()[1])
