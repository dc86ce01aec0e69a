#!/bin/bash
# second half of the evidence set (tools/evidence_a.sh is the first): PMC counters, one rocprofv3 --pmc pass per counter
# group (never combined with tracing), for the headline in both types and the q1422 batch in both types
set -e
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tools/pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
echo "pmc headline f64 done"
tools/pmc.sh $OUT/pmc_f32 bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0 > $OUT/pmc_f32.log 2>&1
echo "pmc headline f32 done"
tools/pmc.sh $OUT/pmc_c3 tools/bench_c3.py --steps 2 --warmup 1 > $OUT/pmc_c3.log 2>&1
tools/pmc.sh $OUT/pmc_c5 tools/bench_c3.py --steps 2 --warmup 1 --dtype f32 > $OUT/pmc_c5.log 2>&1
echo "pmc q1422 done"
