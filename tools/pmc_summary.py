#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv for the ADMM kernel: per-wave averages."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(list)
for r in rows:
    if "biconvex" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
waves = sum(agg.get("SQ_WAVES", [1])) / max(len(agg.get("SQ_WAVES", [1])), 1)
print("launches", len(next(iter(agg.values()))), "waves/launch", waves)
for k, v in sorted(agg.items()):
    m = sum(v) / len(v)
    print("%-24s %14.0f per launch  %12.1f per wave" % (k, m, m / waves))
