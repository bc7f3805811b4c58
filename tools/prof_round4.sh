#!/bin/bash
# Judged artefacts of round 4 (GPU box, repo root): tools/prof_round4.sh <tag>
#  1. the driver's bench line (python3 bench.py --steps 20 --warmup 5) under rocprofv3 --kernel-trace --stats
#  2. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes, counters only with --kernel-trace) of the headline kernel and of every
#     kernel of the KinoDyn workloads (Solo12 at num_iters 10 and 100, Go2 H = 60), and of BASELINE config 3's kernels
#  3. SQ counters of the headline kernel and of the IK kernels (LDS bank conflicts, waits)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_$tag.log 2> $R/gpurun_out/bench_$tag.err
grep -a '"metric"' $R/gpurun_out/bench_$tag.log > $R/gpurun_out/bench_$tag.json
echo "bench done rc $?"
for w in biconvex solo12_h20 solo12_n100 go2_h60 go2_bound_f32 go2_bound_f64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_${w}_$c -- python3 $R/tools/pmc_workload.py $w 2 > $R/gpurun_out/pmc_${tag}_${w}_$c.log 2>&1
    echo "pmc $w $c done rc $?"
  done
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- python3 $R/tools/pmc_workload.py biconvex 2 > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_sq > $R/gpurun_out/pmc_${tag}_sq.txt
cat $R/gpurun_out/pmc_${tag}_sq.txt
cd $R && bash tools/run_pmc_ik.sh ikpmc_$tag 4096 > gpurun_out/ikpmc_$tag.txt 2>&1
grep "ik_backward\|ik_calcdiff1" gpurun_out/ikpmc_$tag.txt | cut -c1-300
echo "all done"
