"""profiles/*_pmc_<workload>.txt (FETCH_SIZE / WRITE_SIZE per launch of the sweep kernel,
summarised by scripts/pmc_summary.py from separate rocprofv3 --pmc passes)
-> profiles/traffic.json, the `roofline.traffic` figure bench.py reports.

Correction (MI355X_MICROARCH.md "HBM" + our own calibration on kernels of known
traffic, profiles/r01_pmc_calibration_v4.txt, r01_pmc_requests_calibration_v4.txt):
  * WRITE_SIZE is exact (k_copy / k_fill / k_gather: reported == written bytes).
  * On gfx950 TCC_BUBBLE and TCC_EA0_RDREQ_32B read 0, so FETCH_SIZE = 64 B x
    TCC_EA0_RDREQ.  A coalesced stream issues 128-B requests (k_copy 1 GiB ->
    8.39 M requests) and is therefore reported at exactly 1/2; a random sub-line
    gather issues one 64-B request per gather (k_gather: 6.0 M gathers + index
    stream -> 6.19 M requests) and is reported in full.
  * Near-sequential gathers of 32-B records -- every record in order, or every other one --
    merge into 128-B line requests like a stream (k_gather32 dense / half_dense in
    tools/microbench: 6 M lanes -> FETCH_SIZE = 1/2 x the bytes of the lines touched,
    profiles/r01_pmc_calibration_v8.txt).
  * Hence for a sweep with G RANDOM gathers per launch:  read bytes = 2 * FETCH - 64 * G.
    G = 2 per edge with the caller's factor order (layout flag 256); with the factors of a
    class sorted by their first variable (the default since v7) one of the two gathers per
    edge end is near-sequential, G = 1 per edge; on a graph with locality (Ising grid) the
    gathers merge into line requests, G = 0 and 2 * FETCH is an upper bound.
  * TILED factor order (round 4): every gather falls into an L2-sized window, so a record
    line goes to the fabric once -- as the 64-byte request of the first gather that touches it -- however many
    gathers hit it afterwards; "64 bytes per random gather" no longer describes what the memory system sees.
    Model for it (spec suffix :tiled, n_gather = the gathered record BYTES per launch, R):
    FETCH = streams / 2 + R  =>  read bytes = 2 * (FETCH - R) + R.  Both raw counters stay in the file.
usage: python scripts/collect_traffic.py TAG workload/dtype=file.txt:n_gather ...
A workload whose cycle is several launches (the n-ary classes: k_factor_nary* + k_variable_wide,
each once per cycle) is given as workload/dtype=file.txt:n_gather:cycle -- the counters of every
kernel launched once per cycle are summed.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_counters(path):
    out = {}
    for line in open(path):
        m = re.search(r"\b([A-Z][A-Z0-9_]+_SIZE)\s+.*k_sweep.*mean(?:_KiB)?\s+([0-9.]+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def read_cycle_counters(path):
    """Sum over the kernels that run once per cycle (n > 1 dispatches in the pass)."""
    out = {}
    for line in open(path):
        m = re.search(r"\b([A-Z][A-Z0-9_]+_SIZE)\s+.*\bn\s+(\d+)\s+mean(?:_KiB)?\s+([0-9.]+)", line)
        if m and int(m.group(2)) > 1:
            out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(3))
    return out


def main():
    out_path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        data = json.load(open(out_path))
    except (OSError, ValueError):
        data = {}
    tag = sys.argv[1]
    for spec in sys.argv[2:]:
        key, rest = spec.split("=")
        path, n_gather, *mode = rest.split(":")
        n_gather = int(n_gather)
        c = read_cycle_counters(path) if mode == ["cycle"] else read_counters(path)
        fetch, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        if mode == ["tiled"]:
            read_bytes = 2 * (fetch - n_gather) + n_gather
            data[key] = {
                "fetch_size_bytes": int(fetch), "write_size_bytes": int(write),
                "gathered_record_bytes_per_launch": n_gather,
                "bytes_per_launch": int(read_bytes + write),
                "model": "tiled factor order: 2*(FETCH_SIZE - R) + R + WRITE_SIZE, R = gathered record bytes "
                         "(see scripts/collect_traffic.py)",
                "source": tag, "file": os.path.relpath(path, ROOT),
            }
            continue
        read_bytes = 2 * fetch - 64 * n_gather
        data[key] = {
            "fetch_size_bytes": int(fetch), "write_size_bytes": int(write),
            "random_gathers_per_launch": n_gather,
            "bytes_per_launch": int(read_bytes + write),
            "model": "2*FETCH_SIZE - 64*gathers + WRITE_SIZE (see scripts/collect_traffic.py)",
            "source": tag, "file": os.path.relpath(path, ROOT),
        }
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data, indent=1))


if __name__ == "__main__":
    main()
