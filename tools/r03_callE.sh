set -e
OUT=gpurun_out/r03e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -x -q -k "fp32 or f32 or config5 or humlicek" > $OUT/pytest_f32.log 2>&1 || (tail -40 $OUT/pytest_f32.log; exit 1)
tail -2 $OUT/pytest_f32.log
python3 bench.py --dtype f32 --no-cpu-baseline > $OUT/headline_f32_bench.json 2> $OUT/err.txt
python3 bench.py --no-cpu-baseline > $OUT/headline_f64_bench.json 2>> $OUT/err.txt
python3 tools/bench_c3.py --steps 20 --dtype f32 > $OUT/c5_bench.json
python3 - <<'PY'
import json
for f in ("headline_f32_bench.json","headline_f64_bench.json"):
    j=json.load(open("gpurun_out/r03e/"+f)); print(f, j["value"], j["roofline"]["avg_launch_ms"], j["acceptance_fraction"])
print(open("gpurun_out/r03e/c5_bench.json").read())
PY
