#!/bin/bash
# The round's judged artefacts in one go (GPU box): smoke, the default bench line under rocprofv3 --kernel-trace --stats,
# the 2-rank rehearsal of the N > 1 path.  usage: tools/prof_final.sh <tag>
tag=$1
set -e
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || { tail -20 gpurun_out/smoke_$tag.log; exit 1; }
grep "smoke ok" gpurun_out/smoke_$tag.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python bench.py > gpurun_out/bench_$tag.log 2>&1
grep -a '"metric"' gpurun_out/bench_$tag.log > gpurun_out/bench_$tag.json
python - <<PY
import csv, glob, json
d = json.load(open("gpurun_out/bench_$tag.json"))
print({k: d[k] for k in ("value", "ms_per_step", "speedup_vs_cpu_all_cores", "p50_latency_ms_batch1", "host_buffers_solves_per_s")})
print(d["roofline"])
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r["Name"][:60], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
BUNMPC_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench2_$tag.log 2>&1
grep -a '"metric"' gpurun_out/bench2_$tag.log | cut -c1-400
