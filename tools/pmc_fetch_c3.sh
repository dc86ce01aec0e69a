#!/bin/bash
# HBM-side traffic of a q1422 half-step (configs 3 / 5) for several builds of the library: FETCH_SIZE and WRITE_SIZE in
# their own rocprofv3 --pmc passes (never combined with tracing), summed over the half-step's kernels.
#   tools/pmc_fetch_c3.sh OUTDIR "bench_c3 args" lib1.so lib2.so ...
set -e
OUT=$1; shift
ARGS=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  name=$(basename $lib .so)
  export VAMP_HIP_LIB=$GRAFT_REPO_ROOT/$lib
  for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --output-format csv -d $OUT/${name}_$tag -- python3 tools/bench_c3.py --steps 2 --warmup 1 $ARGS > $OUT/${name}_$tag.log 2>&1 || echo "$name $grp failed"
  done
  echo "pmc $name done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os, re
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob(out+"/*/**/*counter_collection.csv", recursive=True):
    name=os.path.relpath(f,out).split(os.sep)[0].rsplit("_",2)[0] if False else os.path.relpath(f,out).split(os.sep)[0]
    name=re.sub(r"_(FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum)$","",name)
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "k_half_step" not in k and "k_draws" not in k: continue
        m=re.search(r"(k_half_step<[^>]*Pack<[^>]*>|k_draws)", k)
        agg[name][m.group(1) if m else k[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for b in sorted(agg):
        tot=collections.defaultdict(float)
        for k in sorted(agg[b]):
            for c,v in sorted(agg[b][k].items()):
                m=sum(v)/len(v); tot[c]+=m
                line=f"{b:12s} {k[:70]:70s} {c:14s} n={len(v):3d} mean per launch={m:.6g}"
                print(line); fh.write(line+"\n")
        line=f"{b:12s} per half-step (one launch of every class + k_draws): "+", ".join(f"{c}={v:.6g}" for c,v in sorted(tot.items()))
        if "FETCH_SIZE" in tot: line+=f"  -> HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2*tot['FETCH_SIZE']+tot.get('WRITE_SIZE',0))*1024/1e9:.3f} GB"
        print(line); fh.write(line+"\n")
PY
