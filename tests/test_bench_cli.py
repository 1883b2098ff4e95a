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
    env = dict(os.environ, MAXSUM_HIP_LIB=build(), MAXSUM_COLLECTIVE=collective,
               MAXSUM_RCCL_LIB=build_fake_rccl(), FAKE_RCCL_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--backend", "gloo", "--vars-per-gpu", "1500"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only, one line
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["scaling"] == "weak" and out["unit"] == "iterations/s" and out["dtype"] == "f64"
    assert out["config"]["n_vars"] == 3000 and f"exchange: {collective}" in out["config"]["parallelism"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert out["config"]["check"] == {"cycles": 8 if collective != "p2p" else out["config"]["check"]["cycles"],
                                      "identical_to_single_engine": True, "differences": 0}


def test_bench_rejects_mismatched_world():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=dict(os.environ, WORLD_SIZE="1"), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
