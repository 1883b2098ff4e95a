"""The plug-in's proxies and session on the REAL libmaxsum_hip.so (MI355X), driven through the
pyDCOP stand-in of tests/standin -- in a child process, because the stand-in must not share a
process with the real pyDCOP that tests/test_gpu_vs_reference.py and friends import.  The cases
live in tests/plugin_standin_cases.py; tests/test_plugin_standin.py is the CPU twin on the emulated
engine; tests/test_gpu_reference_e2e.py is the same plug-in under the real orchestrator."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import plugin_standin_cases as T
mod = T.load_plugin()
from oracle import maxsum_oracle
maxsum_oracle.build()
T.test_native_library_is_the_hip_build(mod)
T.test_graph_coloring1_through_the_proxies(mod, "f64")
T.test_graph_coloring1_through_the_proxies(mod, "f32")
T.test_random_coloring_through_the_proxies_equals_oracle(mod, maxsum_oracle)
T.test_change_factor_function_and_stop_of_one_proxy(mod)
T.test_amaxsum_gpu_through_the_proxies_equals_oracle(mod, maxsum_oracle)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "maxsum_hip" in l or "maxsum_emu" in l})
print("LOADED", libs)
print("STANDIN-OK")
'''


def test_plugin_proxies_and_session_on_the_hip_library():
    r = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "STANDIN-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    loaded = [l for l in r.stdout.splitlines() if l.startswith("LOADED")][-1]
    assert "pydcop_amd/csrc/libmaxsum_hip.so" in loaded and "emu" not in loaded, loaded
