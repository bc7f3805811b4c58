#!/usr/bin/env python3
"""kernel trace (rocprofv3 --kernel-trace csv) -> timeline of the LAST IK batch solve: every select / fused launch and the
per-iteration lock-step kernels (start time relative to ik_init, duration, workgroups)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
init = [i for i, r in enumerate(rows) if "ik_init_kernel" in r["Kernel_Name"]]
seg = rows[init[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    n = r["Kernel_Name"]
    short = next((k for k in ("ik_state", "ik_calcdiff", "ik_backward", "ik_forward", "ik_select", "ik_fused", "ik_publish") if k in n), None)
    if short is None:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    if short in ("ik_select", "ik_fused") or d > 30:
        print("%8.3f ms  %-12s wg %5d  %9.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e6, short, wg, d))
