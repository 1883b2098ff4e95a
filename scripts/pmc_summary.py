"""Summarise a rocprofv3 counter_collection CSV: per (kernel, grid) the number of
dispatches and the mean / last counter value (FETCH_SIZE and WRITE_SIZE are in KiB).
usage: python scripts/pmc_summary.py counter_collection.csv COUNTER_NAME"""
import collections
import csv
import sys

acc = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    k = (row["Kernel_Name"][:48], row["Grid_Size"])
    acc.setdefault(k, []).append(float(row["Counter_Value"]))
for (k, g), v in acc.items():
    print(f"{sys.argv[2]} {k:48s} grid {g:>10s} dispatches {len(v):5d} "
          f"mean_KiB {sum(v) / len(v):12.1f} last_KiB {v[-1]:12.1f}")
