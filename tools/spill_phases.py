#!/usr/bin/env python3
"""scratch stores / loads of the one-wave Riccati kernel per phase of its node loop (a -DBWD_MARK build: the PSTAMPV points leave
'; BWDMARK k' comments in the ISA).  usage: python tools/spill_phases.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-DBWD_MARK", "-S", "--cuda-device-only",
       "-I" + ROOT + "/include", ROOT + "/bunmpc_amd/csrc/ik_ddp.hip", "-o", "/tmp/ik_marks.s"] + sys.argv[1:]
subprocess.run(cmd, stderr=subprocess.DEVNULL, cwd="/tmp")
s = open("/tmp/ik_marks.s").read().split("\n")
name = "ik_backward_kernelILi1E"
start = next(i for i, l in enumerate(s) if name in l and ":" in l and not l.startswith("\t") and not l.startswith("."))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
cur, stats, seen_loop = "before the node loop", {}, False
for l in s[start:end]:
    m = re.search(r"BWDMARK (\d+)", l)
    if m:
        cur = "after mark " + m.group(1)
        continue
    d = stats.setdefault(cur, dict(S=0, L=0, n=0))
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith(";") and not t.startswith("."):
        d["n"] += 1
    d["S"] += "scratch_store" in l
    d["L"] += "scratch_load" in l
tot = [0, 0]
for k, v in stats.items():
    print("%-24s instructions %5d  scratch stores %3d loads %3d" % (k, v["n"], v["S"], v["L"]))
    if k != "before the node loop":
        tot[0] += v["S"]; tot[1] += v["L"]
print("inside the node loop (marks 0..8; the segment after mark 8 wraps to the loop top): stores %d loads %d" % tuple(tot))
