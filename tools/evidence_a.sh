set -e
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
python3 bench.py --dtype f32 --no-cpu-baseline --sustain-seconds 2.5 > $OUT/headline_f32_bench.json 2>> $OUT/bench.err
python3 bench.py --force-dist --no-cpu-baseline --sustain-seconds 2.5 > $OUT/bench_rccl_rehearsal_world1.json 2>> $OUT/bench.err
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
python3 tools/fit_q1422.py --quiet --profile $OUT/fit_profile_f64.txt --dump $OUT/fit_f64.npz > $OUT/fit_f64.json 2> $OUT/fit.err
python3 tools/fit_q1422.py --quiet --dtype f32 --dump $OUT/fit_f32.npz > $OUT/fit_f32.json 2>> $OUT/fit.err
python3 tools/robustness.py --steps 10 > $OUT/robustness.txt 2> $OUT/robustness.err || echo "robustness failed"
echo "all done"
