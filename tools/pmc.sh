#!/bin/bash
# Collect PMC counters of the half-step kernel on the GPU box, one rocprofv3 pass per counter group
# (never combined with tracing).  usage: tools/pmc.sh OUTDIR [program args...]
#   default program: bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0
#   e.g. tools/pmc.sh gpurun_out/pmc_c3 tools/bench_c3.py --steps 2 --warmup 1
set -e
OUT=$1; shift
if [ $# -eq 0 ]; then set -- bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
  echo "pmc group $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "k_half_step" not in k and "k_draws" not in k: continue
        m=re.search(r"(k_half_step<[^>]*Pack<[^>]*>|k_draws)", k)
        agg[m.group(1) if m else k[:120]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out+"/summary.txt","w") as fh:
    for k in sorted(agg):
        fh.write("# kernel: "+k+"\n"); print("# kernel:",k)
        for c,vals in sorted(agg[k].items()):
            line=f"{c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}"
            print(line); fh.write(line+"\n")
PY
