#!/bin/bash
# PMC counters of k_half_step for several builds of the library (A/B of instruction counts):
#   tools/pmc_ab.sh OUTDIR lib1.so lib2.so ...     (two rocprofv3 --pmc passes per library, bench.py 2 steps)
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  export VAMP_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $OUT/${name}_p$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0 > $OUT/${name}_p$i.log 2>&1 || echo "$name group $i failed"
  done
  echo "pmc $name done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/*_p*/**/*counter_collection.csv", recursive=True):
    name=os.path.relpath(f,out).split(os.sep)[0].rsplit("_p",1)[0]
    for row in csv.DictReader(open(f)):
        if "k_half_step" not in row["Kernel_Name"]: continue
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for k in sorted(agg):
        fh.write("# build: "+k+"\n"); print("# build:",k)
        for c,vals in sorted(agg[k].items()):
            line=f"{c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}"
            print(line); fh.write(line+"\n")
PY
