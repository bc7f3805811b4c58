#!/usr/bin/env python3
"""per dispatch of tools/scratch/lds_conflict_probe: LDS bank-conflict cycles / LDS-active cycles.  usage: lds_probe_summary.py <rocprofv3 dir> <probe stdout>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
names = [l.split(": ", 1)[1].strip() for l in open(sys.argv[2]) if l.startswith("dispatch ")]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "probe" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k, (d, c) in enumerate(sorted(per.items())):
    a = c.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0
    print("%-64s conflict %8.0f idx_active %8.0f ratio %.2f  insts %6.0f  active_inst_lds %8.0f" % (names[k] if k < len(names) else "?", c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0),
          c.get("SQ_LDS_BANK_CONFLICT", 0) / a, c.get("SQ_INSTS_LDS", 0), c.get("SQ_ACTIVE_INST_LDS", 0)))
