"""Cycles per second of the DSA and MGM engines (pydcop_amd/csrc/dsa.hip, mgm.hip) on the
100k-variable colouring instance of the bench and on the meeting instance (24 values, arity 3):

    python tools/local_search_bench.py [--cycles 500]

One JSON line per (algorithm, instance, kernels); "kernels": "packed" = the default (lane per
constraint where the instance allows it, local_search.h), "slots" = the thread-per-variable
register-array kernels on the slot view (MAXSUM_LOCAL_SEARCH_GENERIC=2), "csr_walk" = the generic
kernels (=1), "strided" = the default kernels without the private row copies of the variables the pack
cannot take (MAXSUM_LOCAL_SEARCH_ROWS=0: their D entries per constraint a stride apart, round 3).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from pydcop_amd import generators as G  # noqa: E402
from pydcop_amd.dsa import DsaEngine  # noqa: E402
from pydcop_amd.graph import Params  # noqa: E402
from pydcop_amd.mgm import MgmEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=500)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--instances", nargs="*", default=["coloring_100k", "meeting_50k"])
    ap.add_argument("--kernels", nargs="*", default=["packed", "strided", "slots", "csr_walk"])
    a = ap.parse_args()
    instances = [("coloring_100k", lambda: G.random_coloring(100_000, seed=0, names=False), Params()),
                 ("meeting_50k", lambda: G.meeting_like(50_000, dom=24, seed=0, names=False), Params(mode="max"))]
    for inst, g, p in instances:
        if inst not in a.instances:
            continue
        g = g()
        for kernels in a.kernels:
            os.environ["MAXSUM_LOCAL_SEARCH_GENERIC"] = {"packed": "0", "strided": "0", "slots": "2", "csr_walk": "1"}[kernels]
            os.environ.pop("MAXSUM_LOCAL_SEARCH_ROWS", None)
            if kernels == "strided":
                os.environ["MAXSUM_LOCAL_SEARCH_ROWS"] = "0"
            for name, make in (("dsa_B", lambda: DsaEngine(g, p, variant="B", seed=1, lib_path=a.lib)),
                               ("mgm", lambda: MgmEngine(g, p, lib_path=a.lib))):
                t0 = time.perf_counter()
                eng = make()
                setup_s = time.perf_counter() - t0
                eng.run(20)
                t0 = time.perf_counter()
                eng.run(a.cycles)
                dt = time.perf_counter() - t0
                print(json.dumps({"algo": name, "instance": inst, "kernels": kernels, "n_vars": g.n_vars,
                                  "cycles_per_s": round(a.cycles / dt, 1), "us_per_cycle": round(1e6 * dt / a.cycles, 2),
                                  "cost": eng.eval_cost()[0], "engine_setup_s": round(setup_s, 2)}), flush=True)
                eng.close()


if __name__ == "__main__":
    main()
