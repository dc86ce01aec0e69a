set -e
OUT=gpurun_out/r03i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q -k "fp32 or f32 or config5 or humlicek or split_workgroup" > $OUT/pytest_f32.log 2>&1 || (tail -60 $OUT/pytest_f32.log; exit 1)
tail -2 $OUT/pytest_f32.log
