"""profiles/*pmc*FETCH_SIZE*.csv + *WRITE_SIZE*.csv -> profiles/traffic.json

Per workload/dtype: mean FETCH_SIZE and WRITE_SIZE (KiB per dispatch, rocprofv3
counter_collection) of the dominant sweep kernel over the profiled launches,
converted to bytes.  Calibration of the two counters on this GPU against kernels
of known traffic is in profiles/*pmc_calibration* (tools/microbench copy/read/fill):
see DESIGN.md section "Measured traffic".

usage: python scripts/collect_traffic.py TAG workload/dtype=fetch.csv,write.csv ...
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, kernel_substr="k_sweep"):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def main():
    out_path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        data = json.load(open(out_path))
    except (OSError, ValueError):
        data = {}
    tag = sys.argv[1]
    for spec in sys.argv[2:]:
        key, files = spec.split("=")
        fetch, write = files.split(",")
        fk, nf = mean_counter(fetch)
        wk, nw = mean_counter(write)
        data[key] = {
            "fetch_kib": fk, "write_kib": wk, "dispatches": min(nf, nw),
            "bytes_per_launch": int((fk + wk) * 1024), "source": tag,
            "files": [os.path.relpath(fetch, ROOT), os.path.relpath(write, ROOT)],
        }
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data, indent=1))


if __name__ == "__main__":
    main()
