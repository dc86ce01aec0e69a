set -e
OUT=gpurun_out/r03a
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_f32 -- python3 $R/bench.py --dtype f32 --steps 60 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $R/$OUT/headline_f32_bench_under_rocprof.json 2> $R/$OUT/trace_f32.err
echo "trace f32 headline done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c5 -- python3 $R/tools/bench_c3.py --steps 20 --dtype f32 > $R/$OUT/c5_bench_under_rocprof.json 2> $R/$OUT/trace_c5.err
echo "trace c5 done"
cd $R
python3 bench.py --dtype f32 --no-cpu-baseline > $OUT/headline_f32_bench.json 2> $OUT/bench_f32.err
python3 bench.py --no-cpu-baseline > $OUT/headline_f64_bench.json 2> $OUT/bench_f64.err
echo "bench done"
tools/pmc.sh $OUT/pmc_f32 bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-chain --sustain-seconds 0 > $OUT/pmc_f32.log 2>&1
echo "pmc f32 headline done"
tools/pmc.sh $OUT/pmc_c5 tools/bench_c3.py --steps 2 --warmup 1 --dtype f32 > $OUT/pmc_c5.log 2>&1
echo "pmc c5 done"
