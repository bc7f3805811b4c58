#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python tools/ik_run.py solo12_h20 4096 2>&1 | grep "solve\|kernel ms\|digest" | tail -4
python tools/ik_run.py go2_h60 1024 2>&1 | grep "solve\|kernel ms\|digest" | tail -3
python -m pytest tests/test_ik_gpu.py tests/test_biconvex_gpu.py -m gpu -x -q > gpurun_out/r4_t6.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4_t6.log
bash tools/run_pmc_ik.sh ikpmc_r4a 4096 > gpurun_out/ikpmc_r4a.txt 2>&1; grep "ik_backward\|ik_calcdiff1" gpurun_out/ikpmc_r4a.txt | cut -c1-400
