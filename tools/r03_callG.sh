set -e
OUT=gpurun_out/r03g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/robustness.py --steps 10 --dtype f64 --pmc-dir $GRAFT_REPO_ROOT/$OUT/pmc64 > $OUT/robustness_f64.txt 2> $OUT/rob64.err
cat $OUT/robustness_f64.txt
python3 tools/robustness.py --steps 10 --dtype f32 > $OUT/robustness_f32.txt 2> $OUT/rob32.err
cat $OUT/robustness_f32.txt
