"""gpurun_out/{bench_<tag>.json, prof_<tag>/} (written by tools/prof_final.sh on the GPU box) -> the files under profiles/
the round is judged on.  usage: python tools/collect_profiles.py <tag> <round prefix, e.g. r01>"""
import csv
import json
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
shutil.copy("gpurun_out/bench_%s.json" % tag, "profiles/%s_final_bench.json" % rnd)
bench = json.load(open("gpurun_out/bench_%s.json" % tag))
rows = list(csv.DictReader(open("gpurun_out/prof_%s/run_kernel_stats.csv" % tag)))
keep = [r for r in rows if "bunmpc" in r["Name"] or "copyBuffer" in r["Name"]]      # torch's own kernels have page-long names
with open("profiles/%s_final_bench_full_kernel_stats.csv" % rnd, "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(keep)
trace = list(csv.DictReader(open("gpurun_out/prof_%s/run_kernel_trace.csv" % tag)))
name = "biconvex_admm_kernel<double, 32, 4, false, false>"
h = sorted((r for r in trace if name in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
n_warm, n_timed = bench["warmup"], bench["steps"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in h]
with open("profiles/%s_final_bench_headline_launches.csv" % rnd, "w") as f:
    f.write("launch,role,Kernel_Name,Grid_Size_X,Workgroup_Size_X,Start_Timestamp,End_Timestamp,duration_us\n")
    for i, r in enumerate(h[:n_warm + n_timed]):
        f.write('%d,%s,"%s",%s,%s,%s,%s,%.3f\n' % (i, "warmup" if i < n_warm else "timed", r["Kernel_Name"], r["Grid_Size_X"],
                                                  r["Workgroup_Size_X"], r["Start_Timestamp"], r["End_Timestamp"], d[i]))
timed = d[n_warm:n_warm + n_timed]
print("headline kernel: %d timed launches, mean %.1f us (bench.py's events: %.1f us)" % (len(timed), sum(timed) / len(timed),
                                                                                         bench["roofline"]["kernel_ms"] * 1e3))
