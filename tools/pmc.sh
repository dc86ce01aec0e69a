#!/bin/bash
# Collect PMC counters for the half-step kernel (run on the GPU box). Separate passes per group.
# usage: tools/pmc.sh OUTDIR [bench args]
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain "$@" > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "k_half_step" not in k: continue
        agg[row["Counter_Name"]]["v"].append(float(row["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for c,v in sorted(agg.items()):
        vals=v["v"]; line=f"{c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}"
        print(line); fh.write(line+"\n")
PY
