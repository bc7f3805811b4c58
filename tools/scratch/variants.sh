#!/bin/bash
# usage (GPU box): tools/scratch/variants.sh V1 V2 ...  -- times the library variants bunmpc_amd/variant_<V>.so on the headline, Go2 fp64 / fp32 and batch-1 configs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  cp bunmpc_amd/variant_$v.so bunmpc_amd/libbunmpc_hip.so
  h=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['p50_latency_ms_batch1'])") || exit 1
  g=$(timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config go2_bound --no-cpu --no-latency --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])") || exit 1
  f=$(timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config go2_bound --precision f32 --no-cpu --no-latency --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])") || exit 1
  echo "variant $v rep $rep: headline, batch-1 p50: $h | go2_bound f64 $g | f32 $f"
done
done
