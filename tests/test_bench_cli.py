"""bench.py --gpus 2 under torch.distributed.run, on the CPU: the N > 1 branch of the
script the driver launches on the multi-GPU node (rank/world handling, sharded runner,
barriers, max-over-ranks timing, the one JSON line of rank 0), run with the emulated
engine over gloo -- with torch's all_to_all and with the engine's own exchange (fake
RCCL).  Numbers are meaningless here; the contract of the line is what is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("collective", ["torch", "rccl"])
def test_bench_two_ranks_gloo(collective, tmp_path):
    from emu.build_emu import build, build_fake_rccl
    build()
    env = dict(os.environ, MAXSUM_COLLECTIVE=collective,
               MAXSUM_RCCL_LIB=build_fake_rccl(), FAKE_RCCL_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "emu", "run_emulated.py"), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--backend", "gloo", "--vars-per-gpu", "300"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only, one line
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "extras"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    # N > 1 = STRONG scaling of ONE fixed instance, BASELINE configs[3] (here scaled down 1 : 333): value is the rate of
    # that instance -- no multiplication by N (VERDICT r4 / ADVICE r3: the weak-scaling aggregate is a labelled extra)
    assert out["scaling"] == "strong" and out["unit"].startswith("iterations/s of ONE 3000-variable instance") and out["dtype"] == "f64"
    cfg = out["config"]
    its = cfg["iterations_per_s_of_this_instance"]
    assert out["value"] == its == out["iterations_per_s_of_the_instance"]
    assert its > 0 and abs(its - 1e3 / out["ms_per_step"]) < 1e-6 * its
    ns = out["north_star_speedup"]   # north_star's ">= 6x at 8 GPUs" reading, at the top level
    assert ns["workload"].startswith("coloring_1m_deg6") and ns["scaling"] == "strong" and ns["speedup_vs_one_gpu"] > 0
    assert abs(ns["speedup_vs_one_gpu"] - out["value"] / out["one_gpu_iterations_per_s"]) < 1e-9
    assert cfg["workload"].startswith("coloring_1m_deg6 (BASELINE configs[3]): strong scaling")
    assert cfg["n_vars"] == 3000 and cfg["n_factors"] == 9000
    assert f"exchange: {collective}" in cfg["parallelism"]
    assert cfg["one_gpu_iterations_per_s"] > 0 and cfg["speedup_vs_one_gpu_on_this_instance"] > 0
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert cfg["check"] == {"cycles": 8, "identical_to_single_engine": True, "differences": 0}
    # labelled extras: the weak-scaling aggregate of the metric's family (ONE 2 x 300-variable instance) and the metric's
    # own instance split two ways
    weak, small = out["extras"]
    assert weak["scaling"] == "weak" and weak["n_vars"] == 600 and weak["workload"] == "coloring_100k x2"
    assert abs(weak["aggregate_iterations_per_s_per_100k_variables"] - 2 * weak["iterations_per_s_of_this_instance"]) < 1e-9 * weak["iterations_per_s_of_this_instance"]
    assert small["scaling"] == "strong" and small["n_vars"] == 300 and small["workload"] == "coloring_100k"
    for e in (weak, small):
        assert e["check"]["identical_to_single_engine"] and e["iterations_per_s_of_this_instance"] > 0
        assert e["speedup_vs_one_gpu"] > 0


def test_bench_single_gpu_line_carries_every_config(tmp_path):
    """N = 1 on the emulated engine, sizes scaled down by a patched workload table: the one JSON
    line has the top-level metric, the cpu_baseline object and one record per extra configuration
    with roofline + traffic source + the parity test id."""
    from emu.build_emu import build
    code = (
        "import sys, runpy; sys.argv = ['bench.py', '--steps', '4', '--warmup', '1', '--vars-per-gpu', '400', '--reference-budget', '4']\n"
        f"from pydcop_amd import engine; engine.register_test_engine({build()!r}, make_default=True)\n"
        "import pydcop_amd.generators as G\n"
        "_ising, _meet, _col = G.ising_grid, G.meeting_like, G.random_coloring\n"
        "G.ising_grid = lambda r, c, **k: _ising(12, 12, **k)\n"
        "G.meeting_like = lambda n, **k: _meet(40, **{**k, 'dom': 6})\n"
        "G.random_coloring = lambda n, **k: _col(min(n, 600), **k)\n"
        "_het = G.meeting_hetero\n"
        "G.meeting_hetero = lambda n, **k: _het(40, **{**k, 'doms': (6, 5)})\n"
        "_peav = G.peav_like\n"
        "G.peav_like = lambda *a, **k: _peav(40, 25, slots=10, max_length=4, max_resources_event=4, **k)\n"
        f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["workload"] == "coloring_100k" and out["dtype"] == "f64"
    # the timed region: repetitions of exactly `steps` cycles, the median one is the line's ms_per_step
    tm = out["timing"]
    assert tm["steps_per_repetition"] == out["steps"] == 4 and tm["repeats"] >= 3 and tm["warmup_by_time_steps"] > 0
    assert tm["ms_per_step_min"] <= out["ms_per_step"] == tm["ms_per_step_median"] <= tm["ms_per_step_max"]
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-9 * out["value"]
    from oracle.stage_reference import locate
    cb = out["cpu_baseline"]
    if locate():   # the reference's own thread-agent runtime, timed in this run, leads; the C port is an extra
        assert cb["kind"] == "reference" and cb["value"] > 0 and cb["cores"] >= 1
        # the reference's rate on the benchmarked instance is an extrapolation of a measured sample, labelled so
        assert cb["extrapolated"] is True and cb["sample_measured_here"] is True and cb["sample_n_vars"] == 1000
        assert cb["port"]["kind"] == "port" and cb["port"]["value"] > 0
        assert all(r["n_vars"] == 1000 for r in cb["thread_agents"])
    else:
        assert cb["kind"] == "port" and cb["value"] > 0
    assert out["roofline"]["traffic_source"] is None or "static" in out["roofline"]["traffic_source"]
    # a cache-resident headline never travels alone: the HBM-resident figure of the same kernel beside it
    assert out["roofline"]["resident"] in ("infinity_cache", "hbm")
    assert out["roofline"]["hbm_resident_reference"]["workload"] == "coloring_1m_deg6"
    got = {(c["workload"], c["dtype"]) for c in out["configs"]}
    assert got == {("coloring_100k", "f32"), ("coloring_10k", "f64"), ("coloring_10k", "f32"),
                   ("ising_1024", "f64"), ("ising_1024", "f32"), ("coloring_1m_deg6", "f64"),
                   ("coloring_1m_deg6", "f32"), ("meeting_50k", "f64"), ("meeting_50k", "f32"),
                   ("peav_50k", "f64"), ("peav_50k", "f32"), ("coloring_100k_d8", "f64"), ("coloring_100k_d8", "f32"),
                   ("meeting_50k_float", "f64"), ("meeting_50k_float", "f32"),
                   ("meeting_50k_hetero", "f64"), ("meeting_50k_hetero", "f32")}
    for c in out["configs"]:
        assert c["parity_checked"] is True and c["parity_test"].startswith("tests/test_gpu_parity.py::")
        rf = c["roofline"]
        for key in ("achieved", "frac", "algorithmic_bytes_per_launch", "avg_launch_us", "traffic", "traffic_source"):
            assert key in rf
        assert c["ms_per_step"] > 0 and c["iterations_per_s"] > 0
    # the widened rows: amaxsum, DSA, MGM on the metric's instance, each naming its parity test
    algos = out["algorithms"]
    assert [a["algo"].split()[0] for a in algos] == ["amaxsum", "dsa", "mgm"]
    assert algos[0]["messages"] > 0 and algos[0]["messages_per_s"] > 0
    for a in algos[1:]:
        assert a["cycles_per_s"] > 0 and a["cost_now"] <= a["cost_at_start"]
    for a in algos:   # every widened row with bytes by a stated formula and a fraction of the peak
        assert a["roofline"]["achieved"] > 0 and 0 < a["roofline"]["frac"] and "formula" in a["roofline"]
    for c in out["configs"]:  # no fraction of the peak above 1 at the top level of a roofline object
        if c["roofline"]["bytes_basis"] == "stored":
            assert c["roofline"]["frac"] == c["roofline"]["frac_of_stored_bytes"] and "frac_algorithmic" in c["roofline"]
    for a in algos:
        assert a["parity_test"].startswith("tests/test_gpu_") and os.path.exists(
            os.path.join(ROOT, a["parity_test"].split("::")[0]))


def test_bench_rejects_mismatched_world():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=dict(os.environ, WORLD_SIZE="1"), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
