set -e
OUT=gpurun_out/r03b
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || (tail -50 $OUT/pytest_gpu.log; exit 1)
tail -3 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
cd /tmp
export VAMP_CLASS_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c3_seq -- python3 $R/tools/bench_c3.py --steps 20 > $R/$OUT/c3_seq_bench_under_rocprof.json 2> $R/$OUT/trace_c3_seq.err
unset VAMP_CLASS_STREAMS
echo "trace c3 seq done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_c3 -- python3 $R/tools/bench_c3.py --steps 20 > $R/$OUT/c3_bench_under_rocprof.json 2> $R/$OUT/trace_c3.err
echo "trace c3 done"
cd $R
python3 tools/bench_c3.py --steps 20 > $OUT/c3_bench.json
tools/pmc.sh $OUT/pmc_c3 tools/bench_c3.py --steps 2 --warmup 1 > $OUT/pmc_c3.log 2>&1
echo "pmc c3 done"
