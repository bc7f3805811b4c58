#!/bin/bash
# Judged artefacts of a round (GPU box, repo root): tools/prof_round2.sh <tag>   (rounds 2 and 3)
#  1. default bench line under rocprofv3 --kernel-trace --stats
#  2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, counters only with --kernel-trace) of the headline kernel and of
#     every kernel of the two KinoDyn workloads
#  3. SQ counters of the headline kernel
tag=$1
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python3 $R/bench.py > $R/gpurun_out/bench_$tag.log 2> $R/gpurun_out/bench_$tag.err
grep -a '"metric"' $R/gpurun_out/bench_$tag.log > $R/gpurun_out/bench_$tag.json
echo "bench done"
for w in biconvex solo12_h20 go2_h60; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_${w}_$c -- python3 $R/tools/pmc_workload.py $w 2 > $R/gpurun_out/pmc_${tag}_${w}_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- python3 $R/tools/pmc_workload.py biconvex 2 > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_sq > $R/gpurun_out/pmc_${tag}_sq.txt
cat $R/gpurun_out/pmc_${tag}_sq.txt
