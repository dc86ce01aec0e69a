#!/bin/bash
# q1422 fit (tools/fit_q1422.py, VAMP_FIT_TIMING=1) for the variants in build/variants named on the command line (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VAMP_HIP_LIB=build/variants/lib_$v.so VAMP_FIT_TIMING=1 python tools/fit_q1422.py --quiet > gpurun_out/abfit_$v.txt 2>/dev/null
  python - gpurun_out/abfit_$v.txt $v <<'PY'
import sys, re, json
run = mp = 0.0
for ln in open(sys.argv[1]):
    m = re.search(r"run=([0-9.]+) ms .*map=([0-9.]+) ms", ln)
    if m: run += float(m.group(1)); mp += float(m.group(2))
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%s: sum of rungs: run %.0f ms, map %.0f ms; fit %.2f s, lines %d, median chi2_r %.6f" % (sys.argv[2], run, mp, d["seconds"], d["lines"], d["median_reduced_chi2"]))
PY
  grep vamp_rung gpurun_out/abfit_$v.txt | head -4 | cut -c1-100
done
