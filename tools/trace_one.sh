#!/bin/bash
# kernel trace of one KinoDyn batch solve -> per-iteration table (GPU box, repo root): tools/trace_one.sh solo12_h20|go2_h60 <tag>
cfg=$1; tag=$2
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$tag -o run -- python3 $R/tools/ik_run.py $cfg > $R/gpurun_out/tr_$tag.log 2>&1
python3 $R/tools/trace_ik.py $(ls $R/gpurun_out/tr_$tag/*/run_kernel_trace.csv $R/gpurun_out/tr_$tag/run_kernel_trace.csv 2>/dev/null | head -n 1) > $R/gpurun_out/trace_$tag.txt
