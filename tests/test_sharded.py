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


def test_multilevel_partitioner_quality_and_corner_cases():
    """pydcop_amd/csrc/partition.cpp (include/maxsum_partition.h) against the first,
    label-propagation version: fewer cut factors, balanced, deterministic."""
    from pydcop_amd import generators as G
    from pydcop_amd.graph import FlatGraph
    g = G.random_coloring(20_000, avg_degree=4, seed=3, names=False)
    for k in (2, 8):
        a = partition_variables(g, k)
        b = partition_variables(g, k)
        np.testing.assert_array_equal(a, b)  # every rank computes it on its own
        sa, sb = cut_statistics(g, a), cut_statistics(g, partition_variables(g, k, method="labelprop"))
        assert set(np.unique(a)) == set(range(k))
        assert sa["edge_imbalance"] < 1.06
        assert sa["cut_factors"] < 0.85 * sb["cut_factors"]
    assert cut_statistics(g, partition_variables(g, 2))["cut_fraction"] < 0.2
    # a grid has a small separator
    gi = G.ising_grid(64, 64, seed=0, names=False)
    assert cut_statistics(gi, partition_variables(gi, 4))["cut_fraction"] < 0.06
    # isolated variables, one hub in every factor (star), an arity-9 factor, more parts than variables
    n = 400
    rowptr = np.arange(0, 2 * (n - 1) + 1, 2, dtype=np.int32)
    ev = np.stack([np.zeros(n - 1, dtype=np.int32), np.arange(1, n, dtype=np.int32)], axis=1).reshape(-1)
    class Star:  # only what partition_variables reads
        n_vars, n_factors, factor_rowptr, edge_var = n + 50, n - 1, rowptr, ev
    p = partition_variables(Star, 4)
    assert p.shape == (n + 50,) and p.min() == 0 and p.max() == 3
    sizes = np.bincount(p, minlength=4)
    assert sizes.min() > 60  # leaves and isolated variables are spread, the hub cannot be
    class Wide:
        n_vars, n_factors = 30, 3
        factor_rowptr = np.array([0, 9, 18, 27], dtype=np.int32)
        edge_var = np.arange(27, dtype=np.int32)
    p = partition_variables(Wide, 3)
    assert sorted(np.bincount(p, minlength=3).tolist()) == [10, 10, 10] or p.max() == 2
    class Tiny:
        n_vars, n_factors = 3, 1
        factor_rowptr = np.array([0, 2], dtype=np.int32)
        edge_var = np.array([0, 1], dtype=np.int32)
    p = partition_variables(Tiny, 8)
    assert p.shape == (3,) and p.min() >= 0 and p.max() < 8


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


@pytest.mark.parametrize("case", ["coloring", "mixed_max", "ising", "coloring_deg9", "scalefree", "secp", "multi"])
@pytest.mark.parametrize("k", [2, 5])
def test_local_shards_equal_single_engine_emu(case, k, emu_lib):
    g, kw = make_case(case)
    _check_against_single(g, kw, k, emu_lib)


def test_fused_sharded_launch_is_used_and_optional(emu_lib, monkeypatch):
    """A shard sweeps a cycle in ONE launch whose last blocks (the cut factor classes) wait
    for the halo inside the kernel; MAXSUM_SHARD_FUSED=0 keeps the two-launch schedule.
    Same results either way."""
    g, kw = make_case("coloring")
    part = partition_variables(g, 3)
    s = build_shard(g, part, 1, 3)
    launches = {}
    for mode in ("1", "0"):  # opt-in: measured slower than two launches on the GPU
        monkeypatch.setenv("MAXSUM_SHARD_FUSED", mode)
        e = MaxSumEngine(s.graph, Params(**kw), lib_path=emu_lib)
        before = e.cycle_bytes()[1]
        e.halo_setup(s.send_edges, s.recv_edges)
        launches[mode] = (before, e.cycle_bytes()[1])
        assert e.shard_mode() == {"fused_launch": mode == "1", "direct_exchange": False, "peer_stores": False}
        e.close()
        _check_against_single(g, kw, 3, emu_lib, steps=(1, 4, 11))
    assert launches["1"] == (2, 1) and launches["0"] == (2, 2)


@pytest.mark.parametrize("flags", [512, 1024, 65536, 65536 + 512])
def test_shard_launch_split_variants_emu(flags, emu_lib):
    """layout_flags 512 / 1024: every factor class / only the cut factor classes in the
    second launch of a sharded cycle -- a scheduling choice, same results.  65536: a cut binary
    factor's replica computes BOTH messages (round 3) instead of only the one to its own variable
    (the default since round 4: the other one goes to a ghost variable nobody sweeps)."""
    for case in ("coloring", "mixed_max"):
        g, kw = make_case(case)
        _check_against_single(g, dict(kw, layout_flags=flags), 3, emu_lib, steps=(1, 5))


def test_shards_in_tiled_factor_order_emu(emu_lib, monkeypatch):
    """The binary factors -- also the cut classes, split by owned position -- in tiled order (layout.cpp; 8 KB
    windows so that these small shards are cut into many tiles): shards == the single engine."""
    monkeypatch.setenv("MAXSUM_TILE_KB", "8")
    for case, k in (("coloring", 3), ("coloring_deg9", 2), ("ising", 4)):
        g, kw = make_case(case)
        _check_against_single(g, kw, k, emu_lib, steps=(1, 5))


def test_local_shards_parity_cases_emu(emu_lib):
    for name, make, kw in parity_cases()[:4] + parity_cases()[9:11]:
        _check_against_single(make(), kw, 3, emu_lib, steps=(1, 6))


def _run_ranks(world, lib, case, steps, tmp_path, timeout=600, extra_env=None):
    port = _free_port()
    out = str(tmp_path / "sharded.npz")
    procs = []
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.update(extra_env or {})
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


@pytest.mark.parametrize("case,world", [("coloring", 2), ("mixed_max", 2), ("coloring", 4)])
def test_gloo_two_ranks_equal_single_engine(case, world, emu_lib, tmp_path):
    """pydcop_amd.sharded.ShardedMaxSum over torch.distributed (gloo, world 2 and 4: a process per rank)."""
    steps = [1, 5, 14]
    z = _run_ranks(world, emu_lib, case, steps, tmp_path)
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


# ---- the native exchange (the engine calls RCCL itself; here: tests/emu/fake_rccl) --------

@pytest.fixture(scope="session")
def fake_rccl():
    from emu.build_emu import build_fake_rccl
    return build_fake_rccl()


@pytest.mark.parametrize("case,k,direct", [("coloring", 3, True), ("mixed_max", 2, False), ("ising", 4, True),
                                           ("coloring", 4, False), ("coloring_deg9", 2, True),
                                           ("coloring_2k", 8, True)])
def test_native_exchange_thread_ranks_emu(case, k, direct, emu_lib, fake_rccl, tmp_path, monkeypatch):
    """mxs_comm_init / mxs_run_sharded: k ranks as k threads of this process, every one
    stepping its own engine through the library's cycle loop."""
    import threading
    from pydcop_amd.engine import comm_unique_id
    monkeypatch.setenv("FAKE_RCCL_DIR", str(tmp_path))
    if case == "coloring" and not direct:
        monkeypatch.setenv("MAXSUM_SHARD_DIRECT", "0")  # pack / unpack kernels, compact buffers
    if case == "coloring_deg9":
        monkeypatch.setenv("MAXSUM_SHARD_FUSED", "1")   # direct exchange + fused launch
    g, kw = make_case(case)
    p = Params(**kw)
    part = partition_variables(g, k)
    shards = [build_shard(g, part, r, k) for r in range(k)]
    uid = comm_unique_id(emu_lib, fake_rccl)
    steps = (1, 2, 9)
    results, errors = [None] * k, []
    modes = [None] * k

    def rank_main(r):
        try:
            s = shards[r]
            e = MaxSumEngine(s.graph, p, lib_path=emu_lib)
            e.halo_setup(s.send_edges, s.recv_edges)
            e.comm_init(r, k, uid, s.send_counts, s.recv_counts, rccl=fake_rccl)
            modes[r] = e.shard_mode()["direct_exchange"]
            assert not e.shard_mode()["peer_stores"]
            e.comm_exchange()
            e.step_unpack()
            out = []
            for n in steps:
                e.run_sharded(n)
                e.sync()
                out.append(e.assignment())
            assert e.cycle_count == sum(steps)
            results[r] = out
            e.close()
        except Exception as ex:  # surfaced in the main thread
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(k)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    # direct exchange (no pack / unpack kernel) wherever the shard qualifies: uniform packed
    # variable classes on the sending side; mixed domains fall back to the staging kernels
    assert modes == [direct] * k, modes
    one = MaxSumEngine(g, p, lib_path=emu_lib)
    for i, n in enumerate(steps):
        one.run(n)
        i1, b1 = one.assignment()
        for r, s in enumerate(shards):
            np.testing.assert_array_equal(results[r][i][0][:s.n_owned], i1[s.local_vars[:s.n_owned]])
            np.testing.assert_array_equal(results[r][i][1][:s.n_owned], b1[s.local_vars[:s.n_owned]])
    one.close()


@pytest.mark.parametrize("case,k,dtype", [("coloring", 3, "f64"), ("ising", 4, "f64"), ("coloring_deg9", 2, "f64"),
                                          ("coloring", 8, "f32"), ("coloring_2k", 8, "f64")])
def test_peer_store_exchange_thread_ranks_emu(case, k, dtype, emu_lib):
    """mxs_peer_export / mxs_peer_connect: no collective -- the variable kernel of a rank stores
    cut-edge records straight into the ghost regions of the others ("IPC" between the rank
    threads of this process), a cycle is one fused launch.  The emulated kernels run one at a
    time, so the ranks move in lockstep here (a barrier per cycle) and never have to poll."""
    import threading
    from pydcop_amd.engine import peer_qualifies
    g, kw = make_case(case)
    p = Params(dtype=dtype, **kw)
    part = partition_variables(g, k)
    shards = [build_shard(g, part, r, k) for r in range(k)]
    steps = (1, 2, 9)
    results, errors, infos, modes = [None] * k, [], [None] * k, [None] * k
    barrier = threading.Barrier(k)

    def rank_main(r):
        try:
            s = shards[r]
            e = MaxSumEngine(s.graph, p, lib_path=emu_lib)
            e.halo_setup(s.send_edges, s.recv_edges)
            infos[r] = e.peer_export(r, k, s.send_counts, s.recv_counts)
            barrier.wait()
            assert all(peer_qualifies(i) for i in infos)
            e.peer_connect(infos)
            modes[r] = e.shard_mode()
            barrier.wait()
            out = []
            for n in steps:
                for _ in range(n):
                    e.run_sharded(1)
                    e.sync()
                    barrier.wait()
                out.append(e.assignment())
            # a reset in the middle of the epochs, then the same cycles again
            barrier.wait()
            e.reset()
            barrier.wait()
            for _ in range(steps[0]):
                e.run_sharded(1)
                e.sync()
                barrier.wait()
            out.append(e.assignment())
            v2f = e.messages()[0]
            assert np.isfinite(v2f).all()
            results[r] = out
            e.close()
        except Exception as ex:
            errors.append((r, repr(ex)))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(k)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(m == {"fused_launch": True, "direct_exchange": False, "peer_stores": True} for m in modes), modes
    one = MaxSumEngine(g, p, lib_path=emu_lib)
    firsts = None
    for i, n in enumerate(steps):
        one.run(n)
        i1, b1 = one.assignment()
        if i == 0:
            firsts = (i1.copy(), b1.copy())
        for r, s in enumerate(shards):
            np.testing.assert_array_equal(results[r][i][0][:s.n_owned], i1[s.local_vars[:s.n_owned]])
            np.testing.assert_array_equal(results[r][i][1][:s.n_owned], b1[s.local_vars[:s.n_owned]])
    for r, s in enumerate(shards):  # after the reset
        np.testing.assert_array_equal(results[r][-1][0][:s.n_owned], firsts[0][s.local_vars[:s.n_owned]])
        np.testing.assert_array_equal(results[r][-1][1][:s.n_owned], firsts[1][s.local_vars[:s.n_owned]])
    one.close()


def test_peer_store_needs_binary_cut_factors(emu_lib):
    """Shards with mixed domains / n-ary cut factors do not qualify: the caller falls back."""
    from pydcop_amd.engine import peer_qualifies
    g, kw = make_case("mixed_max")
    part = partition_variables(g, 2)
    s = build_shard(g, part, 0, 2)
    e = MaxSumEngine(s.graph, Params(**kw), lib_path=emu_lib)
    e.halo_setup(s.send_edges, s.recv_edges)
    assert not peer_qualifies(e.peer_export(0, 2, s.send_counts, s.recv_counts))
    assert e.shard_mode()["peer_stores"] is False
    e.close()


def test_native_exchange_rejects_bad_counts(emu_lib, fake_rccl, tmp_path, monkeypatch):
    from pydcop_amd.engine import MaxSumGpuError, comm_unique_id
    monkeypatch.setenv("FAKE_RCCL_DIR", str(tmp_path))
    g, kw = make_case("coloring")
    part = partition_variables(g, 2)
    s = build_shard(g, part, 0, 2)
    e = MaxSumEngine(s.graph, Params(**kw), lib_path=emu_lib)
    uid = comm_unique_id(emu_lib, fake_rccl)
    with pytest.raises(MaxSumGpuError, match="halo_setup"):
        e.comm_init(0, 2, uid, s.send_counts, s.recv_counts, rccl=fake_rccl)
    with pytest.raises(MaxSumGpuError, match="no communicator"):
        e.halo_setup(s.send_edges, s.recv_edges)
        e.run_sharded(1)
    with pytest.raises(MaxSumGpuError, match="counts"):
        e.comm_init(0, 2, uid, s.send_counts + 1, s.recv_counts, rccl=fake_rccl)
    with pytest.raises(MaxSumGpuError, match="cannot load RCCL"):
        e.comm_init(0, 2, uid, s.send_counts, s.recv_counts, rccl=str(tmp_path / "nope.so"))
    e.close()


def test_gloo_two_ranks_native_exchange(emu_lib, fake_rccl, tmp_path):
    """ShardedMaxSum(collective="rccl") end to end in two processes: unique id handed over
    through torch.distributed, communicator created by the engine, cycle loop in the library."""
    steps = [2, 11]
    env = {"MAXSUM_COLLECTIVE": "rccl", "MAXSUM_RCCL_LIB": fake_rccl, "FAKE_RCCL_DIR": str(tmp_path)}
    z = _run_ranks(2, emu_lib, "mixed_max", steps, tmp_path, extra_env=env)
    assert str(z["collective"]) == "rccl"
    g, kw = make_case("mixed_max")
    one = MaxSumEngine(g, Params(**kw), lib_path=emu_lib)
    done = 0
    for n in steps:
        one.run(n)
        done += n
        i1, b1 = one.assignment()
        np.testing.assert_array_equal(z[f"idx_{done}"], i1)
        np.testing.assert_array_equal(z[f"bel_{done}"], b1)


def test_gloo_two_ranks_peer_stores_fall_back_together(emu_lib, tmp_path):
    """collective="p2p" where the peers cannot be mapped (two processes of emulated engines:
    host memory): every rank sees the failed connect, all start over with a fresh engine and the
    collective, and the results are the single engine's."""
    steps = [2, 6]
    z = _run_ranks(2, emu_lib, "coloring", steps, tmp_path, extra_env={"MAXSUM_COLLECTIVE": "p2p"})
    assert str(z["collective"]) == "torch"
    g, kw = make_case("coloring")
    one = MaxSumEngine(g, Params(**kw), lib_path=emu_lib)
    done = 0
    for n in steps:
        one.run(n)
        done += n
        np.testing.assert_array_equal(z[f"idx_{done}"], one.assignment()[0])
        np.testing.assert_array_equal(z[f"bel_{done}"], one.assignment()[1])


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
@pytest.mark.parametrize("case", ["coloring", "coloring_50k"])
def test_peer_stores_two_processes_one_gpu(case, tmp_path):
    """ShardedMaxSum(collective="p2p") in two processes that share the one GPU of the test
    box: real hipIpc handles, stores into the other process's ghost regions, flag words
    polled by the fused launch -- everything of the peer-store exchange except the xGMI hop."""
    steps = [3, 8]
    try:
        z = _run_ranks(2, None, case, steps, tmp_path, timeout=200,
                       extra_env={"MAXSUM_COLLECTIVE": "p2p", "MAXSUM_TEST_BACKEND": "gloo"})
    except AssertionError as e:
        # the in-kernel waits are time-limited; two processes that time-slice ONE GPU instead of
        # running side by side can exceed the limit -- an artefact of this test set-up, not of
        # the exchange (ranks of a real run own a GPU each).  Wrong results still fail below.
        if "for a halo exchange that" in str(e):
            pytest.skip("the two processes did not run concurrently on the shared GPU")
        raise
    assert str(z["collective"]) == "p2p"
    g, kw = make_case(case)
    one = MaxSumEngine(g, Params(**kw))
    done = 0
    for n in steps:
        one.run(n)
        done += n
        i1, b1 = one.assignment()
        np.testing.assert_array_equal(z[f"idx_{done}"], i1)
        np.testing.assert_array_equal(z[f"bel_{done}"], b1)


@pytest.mark.gpu
@pytest.mark.parametrize("collective", ["rccl", "torch"])
def test_nccl_world1_code_path(collective, tmp_path):
    """ShardedMaxSum with backend nccl (RCCL) at world_size 1 (one GPU on the test box).
    rccl: the engine's own communicator (real librccl: ncclCommInitRank, grouped
    ncclSend / ncclRecv on the comm stream, cycle loop in the library); torch: external
    stream + torch-owned halo tensors bound to the engine."""
    z = _run_ranks(1, None, "coloring_50k", [25], tmp_path, timeout=150,
                   extra_env={"MAXSUM_COLLECTIVE": collective})
    assert str(z["collective"]) == collective
    g, kw = make_case("coloring_50k")
    one = MaxSumEngine(g, Params(**kw))
    one.run(25)
    np.testing.assert_array_equal(z["idx_25"], one.assignment()[0])
    np.testing.assert_array_equal(z["bel_25"], one.assignment()[1])


@pytest.mark.parametrize("case,k", [("coloring", 2), ("ising", 3), ("mixed_max", 2), ("coloring_2k", 4), ("coloring_2k", 8),
                                    ("scalefree", 3), ("secp", 2)])
def test_local_sharded_one_process_equals_single_engine(case, k, emu_lib, fake_rccl, tmp_path, monkeypatch):
    """pydcop_amd.sharded.LocalShardedMaxSum (what the plugin's `devices` parameter runs): k
    shards on k (emulated) devices driven by k threads of ONE process through the library's own
    cycle loop and exchange -- bit-identical to one engine, through reset and a table update."""
    from pydcop_amd.sharded import LocalShardedMaxSum
    monkeypatch.setenv("FAKE_RCCL_DIR", str(tmp_path))
    monkeypatch.setenv("EMU_HIP_DEVICES", str(k))
    g, kw = make_case(case)
    p = Params(**kw)
    one = MaxSumEngine(g, p, lib_path=emu_lib)
    with LocalShardedMaxSum(g, p, list(range(k)), lib_path=emu_lib, rccl=fake_rccl) as many:
        assert many.collective == "rccl" and many.world == k
        done = 0
        for n in (0, 1, 2, 6):
            one.run(n), many.run(n)
            done += n
            assert many.cycle_count == one.cycle_count == done
            np.testing.assert_array_equal(many.assignment()[0], one.assignment()[0])
            np.testing.assert_array_equal(many.assignment()[1], one.assignment()[1])
            a, b = many.eval_cost(), one.eval_cost()
            assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-9 * max(1.0, abs(b[0]))
        # change_factor_function on a replicated factor, then carry on
        f = int(np.flatnonzero(np.diff(g.factor_rowptr) == 2)[0])
        t = np.arange(int(g.table_off[f + 1] - g.table_off[f]), dtype=np.float64)[::-1].copy()
        one.update_factor_table(f, t), many.update_factor_table(f, t)
        one.run(3), many.run(3)
        np.testing.assert_array_equal(many.assignment()[0], one.assignment()[0])
        np.testing.assert_array_equal(many.assignment()[1], one.assignment()[1])
        one.reset(), many.reset()
        one.run(4), many.run(4)
        np.testing.assert_array_equal(many.assignment()[1], one.assignment()[1])
    with pytest.raises(ValueError):
        LocalShardedMaxSum(g, p, [0, 0], lib_path=emu_lib, rccl=fake_rccl)
    one.close()


def test_local_sharded_boot_fails_instead_of_hanging(emu_lib, fake_rccl, tmp_path, monkeypatch):
    """One rank failing BEFORE the communicator (engine creation on a device that does not
    exist) must surface as an error in the caller, not leave the other ranks blocked in
    ncclCommInitRank; a rank that never reaches the communicator must end in a timeout error."""
    import time
    from pydcop_amd.engine import MaxSumGpuError
    from pydcop_amd import sharded
    monkeypatch.setenv("FAKE_RCCL_DIR", str(tmp_path))
    monkeypatch.setenv("EMU_HIP_DEVICES", "2")
    g, kw = make_case("coloring")
    t0 = time.monotonic()
    with pytest.raises(MaxSumGpuError, match="rank 1"):
        sharded.LocalShardedMaxSum(g, Params(**kw), [0, 7], lib_path=emu_lib, rccl=fake_rccl)   # device 7: absent
    assert time.monotonic() - t0 < 60
    # a wedged communicator: rank 1 never calls comm_init
    real = MaxSumEngine.comm_init

    def comm_init(self, rank, *a, **k):
        if rank == 1:
            time.sleep(30)
            raise RuntimeError("late")
        return real(self, rank, *a, **k)
    monkeypatch.setattr(MaxSumEngine, "comm_init", comm_init)
    monkeypatch.setattr(sharded.LocalShardedMaxSum, "COMM_TIMEOUT_S", 3.0)
    monkeypatch.setenv("FAKE_RCCL_TIMEOUT_S", "5")
    t0 = time.monotonic()
    with pytest.raises(MaxSumGpuError, match="did not return within|rank 0"):
        sharded.LocalShardedMaxSum(g, Params(**kw), [0, 1], lib_path=emu_lib, rccl=fake_rccl)
    assert time.monotonic() - t0 < 25


def test_eight_way_cut_of_the_degree6_colouring_stays_under_half():
    """north_star's configs[3] (degree-6 random 3-colouring, 8-way cut): the partitioner
    (csrc/partition.cpp, the METIS stand-in) has to keep the exchange volume where DESIGN
    section 6 / profiles/partition_coloring_1m_deg6_k8.json put it -- under half of the factors
    cut (SURVEY 8(e) guessed 60 %), shards within 2 % of each other.  A 200k-variable instance of
    the same family here (the cut fraction of a random graph does not depend on its size); the
    recorded statistics of the 1M instance itself are checked against the same bounds."""
    import json
    from pydcop_amd import generators as G
    from pydcop_amd.partition import cut_statistics, partition_variables
    g = G.random_coloring(200_000, avg_degree=6, n_colors=3, seed=0, names=False)
    st = cut_statistics(g, partition_variables(g, 8))
    assert st["cut_fraction"] < 0.50 and st["edge_imbalance"] < 1.02, st
    rec = json.load(open(os.path.join(ROOT, "profiles", "partition_coloring_1m_deg6_k8.json")))
    assert rec["n_vars"] == 1_000_000 and rec["cut_factor_fraction"] < 0.50 and rec["edge_imbalance"] < 1.02
    assert rec["halo_bytes_per_rank_per_cycle"]["f64"] < 10 << 20    # < 10 MB per rank and cycle over xGMI
