set -e
OUT=gpurun_out/r03d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/variants.py run 20 --dtype f32 --sustain-seconds 0 > $OUT/phase_f32.txt 2>&1
cat $OUT/phase_f32.txt
python3 tools/variants.py runc3 10 --dtype f32 > $OUT/phase_c5.txt 2>&1
cat $OUT/phase_c5.txt
python3 tools/bench_c3.py --steps 20 > $OUT/c3_bench.json
python3 tools/bench_c3.py --steps 20 --dtype f32 > $OUT/c5_bench.json
cat $OUT/c3_bench.json $OUT/c5_bench.json
