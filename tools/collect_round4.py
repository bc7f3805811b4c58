#!/usr/bin/env python3
"""gpurun_out/{bench_<tag>.json, prof_<tag>/, pmc_<tag>_*, ikpmc_<tag>.txt} (tools/prof_round4.sh) -> profiles/<rnd>_* and profiles/pmc_traffic.json.
usage: python tools/collect_round4.py <tag> <rnd, e.g. r04>"""
import collections, csv, glob, json, os, re, shutil, sys


def newest(pattern):
    """a tag used twice leaves the earlier run's files beside the new ones (gpurun merges directories): take the latest"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

tag, rnd = sys.argv[1], sys.argv[2]
bench = json.load(open("gpurun_out/bench_%s.json" % tag))
shutil.copy("gpurun_out/bench_%s.json" % tag, "profiles/%s_bench.json" % rnd)
stats = newest("gpurun_out/prof_%s/**/*kernel_stats.csv" % tag)
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if "bunmpc" in r["Name"] or "copyBuffer" in r["Name"]]
with open("profiles/%s_bench_kernel_stats.csv" % rnd, "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(keep)
trace = list(csv.DictReader(open(newest("gpurun_out/prof_%s/**/*kernel_trace.csv" % tag))))
name = "biconvex_admm_kernel<double, 32, 4, false, false, 2>"      # (the two-waves-per-SIMD build: what the dispatch takes at B = 4096)
h = sorted((r for r in trace if name in r["Kernel_Name"] and int(r["Grid_Size_X"]) == 2048 * 64), key=lambda r: int(r["Start_Timestamp"]))
n_warm, n_timed = bench["warmup"], bench["steps"]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in h]
with open("profiles/%s_bench_headline_launches.csv" % rnd, "w") as f:
    f.write("launch,role,Kernel_Name,Grid_Size_X,Workgroup_Size_X,Start_Timestamp,End_Timestamp,duration_us\n")
    for i, r in enumerate(h[:n_warm + n_timed]):
        f.write('%d,%s,"%s",%s,%s,%s,%s,%.3f\n' % (i, "warmup" if i < n_warm else "timed", r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"],
                                                  r["Start_Timestamp"], r["End_Timestamp"], d[i]))
timed = d[n_warm:n_warm + n_timed]
print("headline kernel: %d timed launches, mean %.1f us (bench.py's events: %.1f us)" % (len(timed), sum(timed) / len(timed), bench["roofline"]["kernel_ms"] * 1e3))

# HBM traffic per solve, per kernel
traffic = json.load(open("profiles/pmc_traffic.json"))
lines = []
keys = {"biconvex": "solo12_trot H=20 B=4096 admm_iters=10 fista_maxit=150 f64",
        "solo12_h20": "kinodyn solo12_h20 H=20 H_ik=10 B=4096 admm_iters=10", "solo12_n100": "kinodyn solo12_h20 H=20 H_ik=10 B=4096 admm_iters=100",
        "go2_h60": "kinodyn go2_h60 H=60 H_ik=30 B=1024 admm_iters=10"}
for w, key in keys.items():
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    solves = 3
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest("gpurun_out/pmc_%s_%s_%s/**/*counter_collection.csv" % (tag, w, c))
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "bunmpc" in r["Kernel_Name"]:
                k = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"]).group(1)
                per[k][c] += float(r["Counter_Value"])
    kern = {}
    tot_f = tot_w = 0.0
    for k, v in sorted(per.items()):
        fk, wk = v["FETCH_SIZE"] / solves, v["WRITE_SIZE"] / solves          # KB per solve
        kern[k] = {"fetch_size_kb_raw": round(fk, 1), "write_size_kb": round(wk, 1), "traffic_bytes": int((2 * fk + wk) * 1024)}
        tot_f += fk
        tot_w += wk
        lines.append("%-12s %-34s FETCH_SIZE %12.1f KB raw (x2 = %12.1f KB)  WRITE_SIZE %12.1f KB   per batch solve" % (w, k[:34], fk, 2 * fk, wk))
    traffic[key] = {"fetch_size_kb_raw": round(tot_f, 1), "write_size_kb": round(tot_w, 1), "traffic_bytes": int((2 * tot_f + tot_w) * 1024),
                    "per_kernel": kern, "raw_log": "%s_pmc_hbm.txt" % rnd,
                    "note": "sums over all launches of one batch solve (3 solves measured, divided by 3); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950"}
    if w == "biconvex":
        k0 = [k for k in kern if "biconvex" in k][0]
        traffic[key].update(kernel="biconvex_admm_kernel<double, 32, 4, false, false, 2>", algorithmic_bytes=37912576, traffic_bytes=kern[k0]["traffic_bytes"],
                            fetch_size_kb_raw=kern[k0]["fetch_size_kb_raw"], write_size_kb=kern[k0]["write_size_kb"])
    lines.append("%-12s TOTAL traffic per batch solve: %.1f MB" % (w, traffic[key]["traffic_bytes"] / 1e6))
open("profiles/%s_pmc_hbm.txt" % rnd, "w").write("\n".join(lines) + "\n")
json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
shutil.copy("gpurun_out/pmc_%s_sq.txt" % tag, "profiles/%s_pmc_sq.txt" % rnd)
# the headline kernel's VALU occupancy from that SQ pass: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave x the waves a SIMD holds (the
# grid of 2048 waves over 1024 SIMDs at the build's two waves per SIMD) -- what bench.py replays as roofline.valu.simd_busy_frac_pmc
sq = {}
for line in open("profiles/%s_pmc_sq.txt" % rnd):
    parts = line.split()
    if len(parts) >= 2 and parts[0].startswith("SQ_"):
        sq[parts[0]] = float(parts[1])
if "SQ_ACTIVE_INST_VALU" in sq and "SQ_WAVE_CYCLES" in sq:
    per_wave = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
    hk = keys["biconvex"]
    traffic[hk].update(valu_active_per_wave_cycle=round(per_wave, 4), waves_per_simd=2, simd_valu_busy_frac=round(min(1.0, 2 * per_wave), 4),
                       valu_insts_per_wave=int(sq.get("SQ_INSTS_VALU", 0) / max(sq.get("SQ_WAVES", 1), 1)), sq_log="%s_pmc_sq.txt" % rnd)
    json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines))
# BASELINE config 3's kernels
cfg3 = []
for w in ("go2_bound_f32", "go2_bound_f64"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest("gpurun_out/pmc_%s_%s_%s/**/*counter_collection.csv" % (tag, w, c))
        tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "biconvex" in r["Kernel_Name"]) / 3
    cfg3.append("%-14s FETCH_SIZE %10.1f KB raw (x2 = %10.1f KB)  WRITE_SIZE %10.1f KB  -> %.1f MB per launch (algorithmic 74.6 MB)"
                % (w, tot["FETCH_SIZE"], 2 * tot["FETCH_SIZE"], tot["WRITE_SIZE"], (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / 1e6))
open("profiles/%s_pmc_hbm_cfg3.txt" % rnd, "w").write("\n".join(cfg3) + "\n")
print("\n".join(cfg3))
# SQ counters of the IK kernels + the LDS conflict ratio the round-3 review asked for
ik = [l for l in open("gpurun_out/ikpmc_%s.txt" % tag).read().splitlines() if l.startswith("ik_")]
def field(line, name):
    m = re.search(name + r"=(\d+)", line)
    return float(m.group(1)) if m else None
ratio = []
for l in ik:
    c, a_ = field(l, "SQ_LDS_BANK_CONFLICT"), field(l, "SQ_ACTIVE_INST_LDS")
    if c is not None and a_:
        w_, wc = field(l, "SQ_WAIT_ANY"), None
        ratio.append("%-14s SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = %.2f" % (l.split()[0], c / a_))
open("profiles/%s_ik_pmc.txt" % rnd, "w").write("rocprofv3 --kernel-trace --pmc (two passes, tools/run_pmc_ik.sh), MI355X; per-launch AVERAGES over one KinoDyn bench leg (Solo12 trot H=20 / H_ik=10, B = 4096, "
    "single stream).  SQ cycle counters are in units of 4 clocks.\n\n" + "\n".join(ik) + "\n\n" + "\n".join(ratio) + "\n")
print("\n".join(ratio))
