"""TEST ONLY: run a script (or `-m module`) with the emulated engine registered as this
process's engine library -- what the CPU tests use where the product would load
libmaxsum_hip.so (subprocess tests: bench.py under torch.distributed.run, the CLIs).

    python tests/emu/run_emulated.py bench.py --gpus 2 ...
    python tests/emu/run_emulated.py -m pydcop_amd.api instance.npz
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def enable():
    from emu.build_emu import build
    from pydcop_amd import engine
    path = build()
    engine.register_test_engine(path, make_default=True)
    return path


if __name__ == "__main__":
    enable()
    args = sys.argv[1:]
    if args[0] == "-m":
        sys.argv = [args[1]] + args[2:]
        runpy.run_module(args[1], run_name="__main__", alter_sys=True)
    else:
        sys.argv = args
        runpy.run_path(args[0], run_name="__main__")
