#!/bin/bash
# total cycles per Riccati node of ik_backward_kernel for each experiment build in tools/variants (B = 1 and 4096)
for f in tools/variants/lib_*.so; do echo "== $f"; BUNMPC_LIB=$PWD/$f python tools/bwd_profile.py 2>&1 | grep "^B"; done
