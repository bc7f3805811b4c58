#!/bin/bash
# usage (GPU box, repo root): tools/run_pmc_ik.sh <name> <kinodyn batch>
# SQ counters of the three IK kernels (per launch averages) for the KinoDyn bench leg; counters only with --kernel-trace.
set -e
name=$1; kb=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_WAVES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d ${out}_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 64 --no-cpu --no-latency --kinodyn-main-only --kinodyn-streams 1 --kinodyn-batch $kb > ${out}_$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("${out}_$tag/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    for k in ("ik_state", "ik_calcdiff1", "ik_calcdiff_", "ik_backward", "ik_forward", "ik_fused"):
        if k in r["Kernel_Name"]:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, " ".join("%s=%.0f" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), "(launches %d)" % len(next(iter(d.values()))))
PY
done
