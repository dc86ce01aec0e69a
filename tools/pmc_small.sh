#!/bin/bash
# PMC counters of a small-ensemble run (tools/bench_c2.py), one rocprofv3 --pmc pass per group, counters only.
# usage (GPU box): tools/pmc_small.sh OUTDIR [bench_c2 args...]
OUT=$1; shift
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  d=$R/$OUT/$(echo $grp | cut -d" " -f1)
  rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/bench_c2.py "$@" > $d.log 2>&1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, os
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:90]
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0])
        acc[k][row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[k][row["Counter_Name"]][1] += 1
for k, v in acc.items():
    print(k)
    for c, (s, n) in sorted(v.items()):
        print(f"   {c:28s} total {s:14.0f} over {n} dispatch rows")
PY
