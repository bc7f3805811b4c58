#!/bin/bash
# HBM traffic of BASELINE config 3's kernels (Go2 bound H = 40, B = 4096; fp32 at two waves per SIMD, fp64 at one): tools/pmc_cfg3.sh <tag>
tag=$1
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in go2_bound_f32 go2_bound_f64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_${w}_$c -- python3 $R/tools/pmc_workload.py $w 2 > $R/gpurun_out/pmc_${tag}_${w}_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
