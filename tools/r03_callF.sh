set -e
OUT=gpurun_out/r03f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/variants.py run 20 --dtype f32 --sustain-seconds 0 > $OUT/phase_f32.txt 2>&1
cat $OUT/phase_f32.txt
