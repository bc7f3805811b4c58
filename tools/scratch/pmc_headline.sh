#!/bin/bash
# SQ counters + HBM traffic of the headline kernel only (GPU box): tools/scratch/pmc_headline.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_sq -- python3 $R/tools/pmc_workload.py biconvex 2 > $R/gpurun_out/pmc_${tag}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_sq > $R/gpurun_out/pmc_${tag}_sq.txt
cat $R/gpurun_out/pmc_${tag}_sq.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_sq2 -- python3 $R/tools/pmc_workload.py biconvex 2 > $R/gpurun_out/pmc_${tag}_sq2.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_sq2 > $R/gpurun_out/pmc_${tag}_sq2.txt
cat $R/gpurun_out/pmc_${tag}_sq2.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_biconvex_$c -- python3 $R/tools/pmc_workload.py biconvex 2 > $R/gpurun_out/pmc_${tag}_biconvex_$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_biconvex_$c
done
