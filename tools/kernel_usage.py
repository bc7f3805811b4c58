#!/usr/bin/env python3
"""one line per kernel of a translation unit: registers, spills, scratch, occupancy, LDS (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_usage.py ik_ddp.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bunmpc_amd import build
src = sys.argv[1]
cmd = [build.HIPCC] + build.FLAGS + build.FILE_FLAGS.get(src, []) + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(build.CSRC, src), "-o", "/tmp/_usage.o"]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (?:Function Name|Name): (\S+)", line) or re.search(r"(?:Function Name|    Name): (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"bunmpc::\(anonymous namespace\)::", "", cur)
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, r in rows.items():
    print("%-46s VGPR %3d AGPR %3d spill %3d scratch %5d occ %d LDS %6d SGPR %3d" % (k[:46], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
          r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1), r.get("SGPRs", -1)))
