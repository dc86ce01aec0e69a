set -e
OUT=gpurun_out/r03c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/variants.py run 20 --dtype f32 --sustain-seconds 0 > $OUT/variants_f32.txt 2>&1
cat $OUT/variants_f32.txt
python3 tools/variants.py run 20 --sustain-seconds 0 > $OUT/variants_f64.txt 2>&1
cat $OUT/variants_f64.txt
