#!/bin/bash
# usage (GPU box, repo root): tools/run_pmc_mem.sh <outdir-name>
# HBM traffic of the ADMM kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), counters only
# with --kernel-trace as gpurun requires.  Units: KB (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide
# coalesced streams by 2x on gfx950; our access pattern is narrow per-lane blocks, so both raw and
# doubled read figures are printed).
set -e
name=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${name}_$c -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --no-kinodyn > $GRAFT_REPO_ROOT/gpurun_out/${name}_$c.log 2>&1
done
python3 - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/${name}_%s/*/*counter_collection.csv" % c)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "biconvex" in r["Kernel_Name"] and r["Counter_Name"] == c]
    print(c, "per launch: %.1f KB over %d launches" % (sum(v) / len(v), len(v)))
PY
