#!/bin/bash
# kernel trace of the two-stream KinoDyn leg (do kernels of the two streams overlap?).  usage (GPU box): tools/prof_two_streams.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -o run -- python bench.py --steps 2 --warmup 1 --no-cpu --no-latency "$@" > gpurun_out/prof_$tag.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
q = collections.Counter(r["Queue_Id"] for r in rows)
print("queues", q)
# overlap: total time during which kernels of >= 2 different queues are running
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1, r["Queue_Id"])); ev.append((int(r["End_Timestamp"]), -1, r["Queue_Id"]))
ev.sort()
run = collections.Counter(); last = None; busy1 = busy2 = 0
for t, d, qid in ev:
    if last is not None:
        n = sum(1 for k, v in run.items() if v > 0)
        if n >= 1: busy1 += t - last
        if n >= 2: busy2 += t - last
    run[qid] += d; last = t
print("time with >=1 queue busy %.1f ms, with >=2 queues busy %.1f ms" % (busy1 / 1e6, busy2 / 1e6))
PY
