#!/bin/bash
# usage: tools/scratch/build_variant.sh NAME file.hip "flags..."  -- compiles ONE source of the library with the given flags (instead of its FILE_FLAGS)
# and links it with the current objects of the other sources into bunmpc_amd/variant_NAME.so (load with BUNMPC_LIB=...)
set -e
cd "$(dirname "$0")/../../bunmpc_amd/csrc"
name=$1; src=$2; flags=$3
key=3a6c9f2bc8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -I../../include -c $src -o /tmp/variant_$name.o
objs=""
for s in biconvex_admm.hip biconvex_admm_f32.hip biconvex_latency.hip bunmpc_capi.hip ik_ddp.hip bunmpc_ik_capi.hip plan_gen.hip id_ctrl.hip perturb.hip; do
  if [ "$s" == "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs _obj/$s.$key.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variant_$name.so $objs
echo built bunmpc_amd/variant_$name.so
