#!/usr/bin/env python3
"""kernel trace (rocprofv3 --kernel-trace csv) -> per-launch durations of the IK kernels of the LAST batch solve, with grid sizes"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
init = [i for i, r in enumerate(rows) if "ik_init_kernel" in r["Kernel_Name"]]
seg = rows[init[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
out, it = [], 0
for r in seg:
    n = r["Kernel_Name"]
    short = "state" if "ik_state" in n else "calcdiff" if "calcdiff" in n else "backward" if "backward" in n else "forward" + n.split("ik_forward_kernel<")[1][0] if "ik_forward" in n else None
    if short is None:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    out.append((short, wg, d, (int(r["Start_Timestamp"]) - t0) / 1e3))
cur = {}
print("iter  state(wg,us)  calcdiff  backward  forward   t_start_ms")
for short, wg, d, ts in out:
    key = short[:7] if short.startswith("forward") else short
    cur[key] = (wg, d, short)
    if key == "forward":
        it += 1
        if it <= 12 or it % 5 == 0:
            print("%3d  %5d %6.1f | %5d %6.1f | %5d %6.1f | %s %5d %6.1f | %.2f" % (it, cur["state"][0], cur["state"][1], cur["calcdiff"][0], cur["calcdiff"][1],
                                                                               cur["backward"][0], cur["backward"][1], cur["forward"][2], wg, d, ts / 1e3))
