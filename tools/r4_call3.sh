#!/bin/bash
# one GPU call: LDS conflict probe under PMC, the fp32 tail diagnosis, HBM traffic of the Solo12 KinoDyn workload
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/ldsprobe -- $R/tools/scratch/lds_conflict_probe > $R/gpurun_out/ldsprobe.out 2> $R/gpurun_out/ldsprobe.err
python3 $R/tools/lds_probe_summary.py $R/gpurun_out/ldsprobe $R/gpurun_out/ldsprobe.out > $R/gpurun_out/ldsprobe.txt 2>&1
cat $R/gpurun_out/ldsprobe.txt
cd $R && python3 tools/fp32_tail.py solo12_trot 1024 > gpurun_out/fp32_tail.txt 2>&1; tail -60 gpurun_out/fp32_tail.txt | cut -c1-200
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r4a_solo12_h20_$c -- python3 $R/tools/pmc_workload.py solo12_h20 2 > $R/gpurun_out/pmc_r4a_solo12_h20_$c.log 2>&1
  echo "pmc $c done"
done
python3 - <<PY
import csv, glob, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob("$R/gpurun_out/pmc_r4a_solo12_h20_%s/**/*counter_collection.csv" % c, recursive=True))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "bunmpc" in r["Kernel_Name"]:
            k = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"]).group(1)
            per[k][c] += float(r["Counter_Value"])
tot = 0
for k, v in sorted(per.items()):
    b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / 3 * 1024
    tot += b
    print("%-40s fetch x2 %10.1f MB  write %10.1f MB" % (k[:40], 2 * v["FETCH_SIZE"] / 3 / 1024, v["WRITE_SIZE"] / 3 / 1024))
print("TOTAL per batch solve %.1f MB" % (tot / 1e6))
PY
