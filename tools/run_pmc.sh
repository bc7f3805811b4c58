#!/bin/bash
# usage (on the GPU box, from the repo root): tools/run_pmc.sh <outdir-name> [bench args...]
# one rocprofv3 PMC pass over bench.py (counters only with --kernel-trace, as gpurun requires)
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-latency "$@" > $out.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out
