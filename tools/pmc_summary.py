#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].split("(")[0].replace("void smac::", "")
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in rows.values() for c in k})
print("kernel," + ",".join(counters) + ",launches")
for name, cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print(name + "," + ",".join(f"{sum(cs[c]) / max(len(cs[c]), 1):.4g}" if c in cs else "" for c in counters) + f",{n}")
