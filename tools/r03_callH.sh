set -e
OUT=gpurun_out/r03h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || (tail -50 $OUT/pytest_gpu.log; exit 1)
tail -3 $OUT/pytest_gpu.log
python3 tools/fit_q1422.py > $OUT/fit_q1422.log 2>&1 || (tail -20 $OUT/fit_q1422.log; exit 1)
tail -1 $OUT/fit_q1422.log
export VAMP_CLASS_STREAMS=0
python3 tools/c3_kernel_times.py $GRAFT_REPO_ROOT/$OUT/c3_phase 10 > $OUT/c3_phase_seq.txt 2>&1
unset VAMP_CLASS_STREAMS
cat $OUT/c3_phase_seq.txt
python3 tools/variants.py run 20 --sustain-seconds 0 > $OUT/headline_variants.txt 2>&1
cat $OUT/headline_variants.txt
