#!/bin/bash
# The evidence set of a build (profiles/README.md lists what each file is), in two GPU-box calls because of the box's
# time limit:   tools/evidence_a.sh   benches, rocprofv3 kernel traces, q1422 fits, robustness table
#               tools/evidence_b.sh   PMC counters, one rocprofv3 --pmc pass per counter group (never combined with tracing)
# both write under gpurun_out/r04_final; run from the repo root on the GPU box.
set -e
bash tools/evidence_a.sh
bash tools/evidence_b.sh
