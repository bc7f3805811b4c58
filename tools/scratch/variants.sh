#!/bin/bash
# usage (GPU box): tools/scratch/variants.sh  -- times the library variants bunmpc_amd/variant_*.so on the headline and fp32 configs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in A B C D; do
  cp bunmpc_amd/variant_$v.so bunmpc_amd/libbunmpc_hip.so
  h=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-latency --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms'))") || exit 1
  f=$(timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config go2_bound --precision f32 --no-cpu --no-latency --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])") || exit 1
  g=$(timeout -k 10 200 python bench.py --steps 20 --warmup 3 --config go2_bound --no-cpu --no-latency --no-kinodyn 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])") || exit 1
  echo "variant $v rep $rep: headline $h | go2_bound f32 $f | go2_bound f64 $g"
done
done
