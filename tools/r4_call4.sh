#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_biconvex_gpu.py tests/test_ik_gpu.py -m gpu -x -q > gpurun_out/r4_t4.log 2>&1; echo "tests rc $?"; grep -a "same discrete path\|passed\|failed" gpurun_out/r4_t4.log | tail -8
for v in "" tools/scratch/libs/lib_wpe2.so; do echo "== lib $v"; BUNMPC_LIB=$v python tools/ik_run.py solo12_h20 4096 2>&1 | grep "solve\|kernel ms\|digest" | tail -4; done
echo "== go2 default lib"; python tools/ik_run.py go2_h60 1024 2>&1 | grep "solve\|kernel ms\|digest" | tail -3
python bench.py --no-cpu --no-kinodyn --steps 10 > gpurun_out/r4_bench4.json 2> gpurun_out/r4_bench4.err; echo "bench rc $?"; python - <<PY
import json
d = json.loads(open("gpurun_out/r4_bench4.json").read().strip().splitlines()[-1])
print("headline ms", d["ms_per_step"], "lanes", d["roofline"]["lanes_per_problem"], "batch_6144", d.get("batch_6144"))
PY
