#!/bin/bash
# One GPU-box call that collects the evidence set of a build (profiles/README.md lists what each file is):
#   tools/evidence.sh OUTDIR      (run from the repo root on the GPU box)
set -e
OUT=$1
mkdir -p $OUT
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/trace.err
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c3 -- python3 $R/tools/bench_c3.py --steps 20 > $R/$OUT/c3_bench_under_rocprof.json 2> $R/$OUT/trace_c3.err
echo "trace c3 done"
cd $R
python3 tools/bench_c3.py --steps 20 > $OUT/c3_bench.json
python3 tools/bench_c3.py --steps 20 --dtype f32 > $OUT/c5_q1422_f32_bench.json
python3 bench.py --dtype f32 --no-cpu-baseline > $OUT/headline_f32_bench.json 2>> $OUT/bench.err
echo "c3/c5 done"
tools/pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
echo "pmc done"
