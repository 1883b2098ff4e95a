"""Random small instances (domains 1..17, arities 1..4, both modes, both precisions, every
start_messages / damping_nodes choice, random layout flags) through an engine build and its oracle,
bit for bit.  `python tests/fuzz_common.py FIRST LAST` runs the seeds FIRST..LAST-1 on the emulated
build; tests/test_fuzz_emu.py runs a few of them in the CPU suite."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pydcop_amd import generators as G  # noqa: E402
from pydcop_amd.graph import Params  # noqa: E402


def instance(seed):
    rng = np.random.default_rng(1000 + seed)
    nv, nf = int(rng.integers(3, 60)), int(rng.integers(1, 120))
    # ($FUZZ_DOMS=big: ad-hoc sweeps over the domains of the round-5 kernels too -- lane grids of 16 / 64 lanes, box overhang)
    choices = [1, 2, 3, 4, 5, 7, 9, 17] if os.environ.get("FUZZ_DOMS") != "big" else [1, 2, 3, 4, 5, 6, 8, 9, 12, 17, 21, 24, 33]
    doms = tuple(int(x) for x in rng.choice(choices, size=int(rng.integers(1, 4))))
    max_arity = int(rng.integers(1, 5))
    if os.environ.get("FUZZ_DOMS") == "big" and max(doms) > 17:
        max_arity = min(max_arity, 3 if max(doms) <= 24 else 2)  # (keeps the emulated sweep's tables small)
    g = G.random_mixed(nv, nf, seed=seed, max_arity=max_arity, dom_choices=doms,
                       float_tables=bool(rng.integers(0, 2)))
    kw = dict(mode="max" if rng.integers(0, 2) else "min", dtype="f32" if rng.integers(0, 3) == 0 else "f64",
              start_messages=["leafs", "leafs_vars", "all"][int(rng.integers(0, 3))],
              damping_nodes=["vars", "factors", "both", "none"][int(rng.integers(0, 4))],
              damping=float(rng.choice([0.0, 0.3, 0.5])), stability=float(rng.choice([0.02, 0.1, 0.5])))
    flags = int(rng.choice([0, 0, 2048, 8192, 256, 16, 8, 2048 | 8192]))
    if os.environ.get("FUZZ_DOMS") == "big":  # (+ the round-5 switches: generic instead of lane grids, no pack8, pack8 in its own launch)
        flags |= int(rng.choice([0, 0, 0, 524288, 1048576, 2097152]))
    return g, kw, flags, rng


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), what


def fuzz_maxsum(seed, lib_path):
    from oracle.maxsum_oracle import OracleMaxSum
    from pydcop_amd.engine import MaxSumEngine
    g, kw, flags, _ = instance(seed)
    with MaxSumEngine(g, Params(layout_flags=flags, **kw), lib_path=lib_path) as e:
        o = OracleMaxSum(g, Params(**kw))
        done = 0
        for n in (0, 1, 2, 5):
            e.run(n), o.run(n)
            done += n
            for a, b in zip(e.messages(), o.messages()):
                assert np.array_equal(a, b), f"messages after {done} cycles"
            _same(e.assignment(), o.assignment(), f"selection after {done} cycles")
        o.close()


def fuzz_others(seed, lib_path):
    from amaxsum_common import same_state
    from oracle.amaxsum_oracle import OracleAMaxSum
    from oracle.dsa_oracle import OracleDsa
    from oracle.mgm_oracle import OracleMgm
    from pydcop_amd.amaxsum import AMaxSumEngine
    from pydcop_amd.dsa import DsaEngine
    from pydcop_amd.mgm import MgmEngine
    g, kw, _, rng = instance(seed)
    e, o = AMaxSumEngine(g, Params(**kw), lib_path=lib_path), OracleAMaxSum(g, Params(**kw))
    for gens in (1, 2, 4, 7):
        if o.pending > 5000:
            break
        assert e.run(gens) == o.run(gens)
        same_state(e, o, f"generations < {gens}")
    e.close(), o.close()
    p = Params(mode=kw["mode"], dtype=kw["dtype"])
    dsa_kw = dict(variant="ABC"[int(rng.integers(0, 3))], probability=float(rng.choice([0.3, 0.7, 1.0])), seed=seed)
    for e, o, n in ((DsaEngine(g, p, lib_path=lib_path, **dsa_kw), OracleDsa(g, p, **dsa_kw), 6),
                    (MgmEngine(g, p, lib_path=lib_path), OracleMgm(g, p), 5)):
        e.run(n), o.run(n)
        _same(e.assignment(), o.assignment(), type(e).__name__)
        e.close(), o.close()


if __name__ == "__main__":
    from emu.build_emu import build
    from pydcop_amd import engine
    from oracle.maxsum_oracle import build as build_oracles
    build_oracles()
    lib, bad = build(), 0
    engine.register_test_engine(lib)
    for s in range(int(sys.argv[1]), int(sys.argv[2])):
        for f in (fuzz_maxsum, fuzz_others):
            try:
                f(s, lib)
            except Exception as ex:  # report and go on
                bad += 1
                print("FAIL", f.__name__, "seed", s, repr(ex)[:300], flush=True)
        if s % 25 == 24:
            print("... seed", s, "failures so far:", bad, flush=True)
    print("failures:", bad)
