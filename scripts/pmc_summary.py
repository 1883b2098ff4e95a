"""Summarise a rocprofv3 counter_collection CSV: per (kernel, grid, counter) the number
of dispatches and the mean / last counter value (FETCH_SIZE and WRITE_SIZE are in KiB).
usage: python scripts/pmc_summary.py counter_collection.csv [label]"""
import collections
import csv
import sys

acc = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    k = (row["Kernel_Name"][:44], row["Grid_Size"], row["Counter_Name"])
    acc.setdefault(k, []).append(float(row["Counter_Value"]))
for (k, g, c), v in acc.items():
    print(f"{c:24s} {k:44s} grid {g:>10s} n {len(v):5d} mean {sum(v) / len(v):14.1f} last {v[-1]:14.1f}")
