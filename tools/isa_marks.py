#!/usr/bin/env python3
"""Where a kernel spills: the sequence of scratch stores (S) / loads (L), MFMAs (M), v_rsq_f64 (R), global loads / stores (G / g),
LDS reads / writes (d / w), readlanes (r) and branches (b) of one function of a hipcc -S listing, run-length coded with line offsets.
usage: python tools/isa_marks.py file.s mangled_name_substring"""
import re, sys
s = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if name in l and not l.startswith("\t") and not l.startswith(".") and ":" in l)
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
body = s[start:end]
print("function at line %d, %d lines" % (start, len(body)))
kinds = [("scratch_store", "S"), ("scratch_load", "L"), ("v_mfma", "M"), ("v_rsq_f64", "R"), ("global_store", "g"), ("global_load", "G"),
         ("ds_read", "d"), ("ds_write", "w"), ("v_readlane", "r"), ("s_cbranch", "b"), ("s_barrier", "B"), ("v_accvgpr", "a")]
marks = []
for i, l in enumerate(body):
    for k, c in kinds:
        if k in l:
            marks.append((i, c))
            break
out, prev, cnt, st = [], None, 0, 0
for i, c in marks:
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append("%s%d@%d" % (prev, cnt, st))
        prev, cnt, st = c, 1, i
out.append("%s%d@%d" % (prev, cnt, st))
print(" ".join(out))
import collections
print(collections.Counter(c for _, c in marks))
