#!/bin/bash
# usage (GPU box): tools/scratch/kd_variants.sh V1 V2 ...  -- KinoDyn legs (Solo12 B=4096, Go2 H=60 B=1024) per library variant ("default" = the built library)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ $v == default ]; then unset BUNMPC_LIB; else export BUNMPC_LIB=$GRAFT_REPO_ROOT/bunmpc_amd/variant_$v.so; fi
  a=$(timeout -k 10 200 python bench.py --workload kinodyn --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d.get('ik_kernel_ms_per_solve',{}).items()})") || exit 1
  g=$(timeout -k 10 200 python bench.py --workload kinodyn --kinodyn-config go2_h60 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d.get('ik_kernel_ms_per_solve',{}).items()})") || exit 1
  echo "variant $v rep $rep: solo12 $a | go2 $g"
done
done
