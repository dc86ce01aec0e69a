#!/bin/bash
# two PMC passes: instruction counts and stall breakdown of the half-step kernel
OUT=$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0 "$@" > $OUT/p$i.log 2>&1 || echo "group $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_half_step" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
ev=32768*262144/64.0
for c,v in sorted(agg.items()):
    m=sum(v)/len(v); print(f"{c:24s} {m:.5g}   per wave-eval {m/ev:.2f}")
PY
