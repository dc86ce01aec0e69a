#!/bin/bash
# the bench lines of tools/evidence_a.sh again, after profiles/pmc_traffic*.json were regenerated from tools/evidence_b.sh's passes
# (the lines quote those files: traffic, VALU instructions)
set -e
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --dtype f32 --no-cpu-baseline --sustain-seconds 2.5 > $OUT/headline_f32_bench.json 2>> $OUT/bench.err
python3 bench.py --force-dist --no-cpu-baseline --sustain-seconds 2.5 > $OUT/bench_rccl_rehearsal_world1.json 2>> $OUT/bench.err
python3 tools/bench_c3.py --steps 20 > $OUT/c3_bench.json
python3 tools/bench_c3.py --steps 20 --dtype f32 > $OUT/c5_q1422_f32_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_f32 -- python3 $R/bench.py --dtype f32 --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/headline_f32_bench_under_rocprof.json 2> $R/$OUT/trace_f32.err
echo done
