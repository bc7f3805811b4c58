#!/usr/bin/env python3
"""kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream): usage tools/trace_gaps.py <kernel_trace.csv> [last N kernels]"""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows = rows[-n:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, r in enumerate(rows):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("bunmpc::(anonymous namespace)::", ""))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if i:
        gap[k].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
tot = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("span of the last %d kernels: %.1f us" % (len(rows), tot))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    g = gap.get(k, [0.0])
    print("%-40s n %5d  duration mean %8.1f us (sum %9.1f)   gap before it: mean %6.2f us (sum %8.1f)" % (k[:40], len(dur[k]), sum(dur[k]) / len(dur[k]), sum(dur[k]), sum(g) / len(g), sum(g)))
