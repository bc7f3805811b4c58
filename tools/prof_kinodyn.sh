#!/bin/bash
# rocprofv3 kernel-trace summary of the KinoDyn bench leg.  usage (on the GPU box): tools/prof_kinodyn.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python bench.py --steps 2 --warmup 1 --no-cpu --no-latency "$@" > gpurun_out/prof_$tag.log 2>&1
grep -a '"metric"' gpurun_out/prof_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["kinodyn_full_solve"]); print("admm ms", d["roofline"]["kernel_ms"])'
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:50], r["Calls"], "total_ms %.2f" % (float(r["TotalDurationNs"])/1e6), "avg_us %.1f" % (float(r["AverageNs"])/1e3), "min_us %.1f" % (float(r["MinNs"])/1e3), "max_us %.1f" % (float(r["MaxNs"])/1e3))
PY
