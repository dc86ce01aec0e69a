#!/bin/bash
# A/B of the variants in build/variants on the small-ensemble cases, config 2, config 3 and the headline (GPU box)
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  export VAMP_HIP_LIB=build/variants/lib_$n.so
  for cfg in "--walkers 32 --comp 1" "--walkers 100 --comp 1" "--walkers 64 --comp 4" "--walkers 4096 --comp 4"; do
    python tools/bench_c2.py $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', d['config'], '%.2f us per half-step' % d['us_per_half_step_wall'])"
  done
done
python tools/variants.py runc3 5 2>&1 | tail -4
python tools/variants.py runc3 5 --dtype f32 2>&1 | tail -4
python tools/variants.py run 10 2>&1 | tail -4
