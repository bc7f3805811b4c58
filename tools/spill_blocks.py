#!/usr/bin/env python3
"""Scratch operations and accumulation-register moves per basic block of one function of a hipcc -S listing (which block of a kernel
pays for its spills; loop bodies show by their size and their back edge).  usage: python tools/spill_blocks.py file.s name_substring [min_instr]"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
min_instr = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(s) if name in l and ":" in l and not l.startswith("\t") and not l.startswith("."))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
blocks, cur = [], dict(label="entry", n=0, ld=0, st=0, acc=0, lds=0, back="")
for l in s[start + 1:end]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = dict(label=t.split(":")[0], n=0, ld=0, st=0, acc=0, lds=0, back="")
        continue
    if not t or t.startswith(";") or t.startswith("."): continue
    cur["n"] += 1
    cur["ld"] += t.startswith("scratch_load")
    cur["st"] += t.startswith("scratch_store")
    cur["acc"] += t.startswith("v_accvgpr")
    cur["lds"] += t.startswith("ds_")
    m = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
    if m:
        tgt = m.group(1) or m.group(2)
        if any(b["label"] == tgt for b in blocks) or tgt == cur["label"]: cur["back"] += " ->" + tgt
blocks.append(cur)
tot = dict(ld=0, st=0, acc=0)
for b in blocks:
    for k in tot: tot[k] += b[k]
    if b["n"] >= min_instr or b["back"]:
        print("%-12s instr %4d  scratch ld %3d st %3d  accvgpr %3d  lds %3d %s" % (b["label"], b["n"], b["ld"], b["st"], b["acc"], b["lds"], b["back"]))
print("total scratch ld %d st %d accvgpr %d over %d blocks" % (tot["ld"], tot["st"], tot["acc"], len(blocks)))
