#!/bin/bash
# third part of the evidence set: the full GPU suite, the soaks, the robustness table and the headline line on the final library
set -e
OUT=gpurun_out/r04_final_c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1
echo "gpu suite done: $(tail -1 $OUT/gputests.log)"
{
  echo "# Round-4 soaks on the final library (GPU box, one call): random regions / samplers / sharded exchange against the oracle, every packing"
  echo "## tests/soak_long_regions.py 200 (fp64)";  python tests/soak_long_regions.py 200
  echo "## tests/soak_long_regions.py 200 f32";     python tests/soak_long_regions.py 200 f32
  echo "## tests/soak_short_regions.py (fp64)";     python tests/soak_short_regions.py 20
  echo "## tests/soak_short_regions.py f32";        python tests/soak_short_regions.py 20 f32
  echo "## tests/soak_sampler.py";                  python tests/soak_sampler.py
  echo "## tests/soak_sharded.py";                  python tests/soak_sharded.py 2>&1 | grep -E "^ok|^soak|FAIL"
  echo "## tests/soak_resident_map.py 120";         python tests/soak_resident_map.py 120
} > $OUT/soaks.txt 2>&1
echo "soaks done"
python tools/robustness.py --steps 10 > $OUT/robustness.txt 2>&1
echo "robustness done"
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
tail -c 600 $OUT/bench.json
