"""CPU twin of tests/test_gpu_plugin.py: the same cases (tests/plugin_standin_cases.py: (plug-in proxies + session driven
through the pyDCOP stand-in of tests/standin) on the emulated engine, in a subprocess so that the
stand-in never shadows the real pyDCOP of the other tests."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
from emu.run_emulated import enable
enable()
import plugin_standin_cases as T
mod = T.load_plugin()
from oracle import maxsum_oracle
maxsum_oracle.build()
T.test_graph_coloring1_through_the_proxies(mod, "f64")
T.test_graph_coloring1_through_the_proxies(mod, "f32")
T.test_random_coloring_through_the_proxies_equals_oracle(mod, maxsum_oracle)
T.test_change_factor_function_and_stop_of_one_proxy(mod)
print("STANDIN-OK")
'''


def test_plugin_through_the_standin_on_the_emulated_engine():
    r = subprocess.run([sys.executable, "-c", CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STANDIN-OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
