#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per-kernel mean of one counter."""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") == counter:
            acc[row.get("Kernel_Name", "?")].append(float(row["Counter_Value"]))
for kname, vals in acc.items():
    print(f"{counter} kernel={kname[:60]} dispatches={len(vals)} mean={sum(vals) / len(vals):.6g} max={max(vals):.6g}")
