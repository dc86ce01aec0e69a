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
python3 bench.py --dtype f32 --no-cpu-baseline > $OUT/headline_f32_bench.json 2>> $OUT/bench.err
python3 bench.py --force-dist --no-cpu-baseline > $OUT/bench_rccl_rehearsal_world1.json 2>> $OUT/bench.err
python3 tools/bench_c3.py --steps 20 > $OUT/c3_bench.json
python3 tools/bench_c3.py --steps 20 --dtype f32 > $OUT/c5_q1422_f32_bench.json
echo "benches done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_f32 -- python3 $R/bench.py --dtype f32 --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/headline_f32_bench_under_rocprof.json 2> $R/$OUT/trace_f32.err
echo "headline traces done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c3 -- python3 $R/tools/bench_c3.py --steps 20 > $R/$OUT/c3_bench_under_rocprof.json 2> $R/$OUT/trace_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c5 -- python3 $R/tools/bench_c3.py --steps 20 --dtype f32 > $R/$OUT/c5_bench_under_rocprof.json 2> $R/$OUT/trace_c5.err
export VAMP_CLASS_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c3_seq -- python3 $R/tools/bench_c3.py --steps 20 > $R/$OUT/c3_seq_bench_under_rocprof.json 2> $R/$OUT/trace_c3_seq.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c5_seq -- python3 $R/tools/bench_c3.py --steps 20 --dtype f32 > $R/$OUT/c5_seq_bench_under_rocprof.json 2> $R/$OUT/trace_c5_seq.err
unset VAMP_CLASS_STREAMS
echo "q1422 traces done"
cd $R
tools/pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
echo "pmc headline f64 done"
tools/pmc.sh $OUT/pmc_f32 bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0 > $OUT/pmc_f32.log 2>&1
echo "pmc headline f32 done"
tools/pmc.sh $OUT/pmc_c3 tools/bench_c3.py --steps 2 --warmup 1 > $OUT/pmc_c3.log 2>&1
tools/pmc.sh $OUT/pmc_c5 tools/bench_c3.py --steps 2 --warmup 1 --dtype f32 > $OUT/pmc_c5.log 2>&1
echo "pmc q1422 done"
