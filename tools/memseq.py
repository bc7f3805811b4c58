#!/usr/bin/env python3
"""The order of memory operations and waits of one function of a hipcc -S listing: L global load, F flat load, s / f stores, r / p scratch
load / store, W a wait on vmcnt, d a run of LDS instructions, | s_barrier, C a call.  A kernel that is waiting most of its cycles shows
as LWLWLW: loads waited for one by one.  usage: python tools/memseq.py file.s function_name_substring"""
import sys
s = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if name in l and ":" in l and not l.startswith("\t") and not l.startswith("."))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
seq = []
for l in s[start:end]:
    t = l.strip()
    if t.startswith("s_waitcnt") and "vmcnt" in t: seq.append("W")
    elif t.startswith("global_load"): seq.append("L")
    elif t.startswith("flat_load"): seq.append("F")
    elif t.startswith("global_store"): seq.append("s")
    elif t.startswith("flat_store"): seq.append("f")
    elif t.startswith("scratch_load"): seq.append("r")
    elif t.startswith("scratch_store"): seq.append("p")
    elif t.startswith("s_barrier"): seq.append("|")
    elif "s_swappc" in t: seq.append("C")
    elif t.startswith("ds_") and (not seq or seq[-1] != "d"): seq.append("d")
print("".join(seq))
