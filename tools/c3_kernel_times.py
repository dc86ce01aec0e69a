#!/usr/bin/env python3
"""Per-kernel average durations of tools/bench_c3.py for every variant library under build/variants
(rocprofv3 kernel trace; run on the GPU box).  usage: python tools/c3_kernel_times.py OUTDIR [steps] [bench_c3 args...]
(VAMP_CLASS_STREAMS=0 in the environment runs the launch classes one after the other: per-class times)"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1])
steps = sys.argv[2] if len(sys.argv) > 2 else "5"
extra = sys.argv[3:]
os.makedirs(out, exist_ok=True)
var = os.path.join(ROOT, "build", "variants")
for lib in sorted(glob.glob(os.path.join(var, "lib_*.so"))):
    name = os.path.basename(lib)[4:-3]
    d = os.path.join(out, name)
    env = dict(os.environ, VAMP_HIP_LIB=lib, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3",
                    os.path.join(ROOT, "tools", "bench_c3.py"), "--steps", steps] + extra, env=env, capture_output=True, cwd="/tmp")
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_half_step<[^>]*Pack<[^>]*>|k_draws)", r["Name"])
            if m:
                rows.append((m.group(1).replace("(anonymous namespace)::", ""), float(r["AverageNs"]) / 1e6))
    print(name, " | ".join(f"{k}: {v:.3f} ms" for k, v in sorted(rows)), flush=True)
