#!/usr/bin/env python3
"""Timing-only builds that force every Voigt evaluation down one branch of voigt_H, to price the
branches on the headline workload.  Build here (no GPU needed):  python tools/tier_cost.py build
Run on the GPU box:                                              python tools/tier_cost.py run
Results are wrong by construction in these builds; only the kernel time is read."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "variants")
SRC = os.path.join(ROOT, "vamp_amd", "csrc", "vamp_hip.hip")
NAMES = {0: "core", 1: "jfrac6", 2: "jfrac4", 3: "jfrac3", 4: "jfrac2", 5: "far", 6: "none"}

if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for t in NAMES:
        so = os.path.join(OUT, f"libvamp_tier{t}.so")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                                       f"-DVAMP_FORCE_TIER={t}", "-o", so, SRC]))
    for p in procs:
        assert p.wait() == 0
else:
    steps = sys.argv[2] if len(sys.argv) > 2 else "3"
    for t, name in NAMES.items():
        env = dict(os.environ, VAMP_HIP_LIB=os.path.join(OUT, f"libvamp_tier{t}.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            ms = j["roofline"]["avg_launch_ms"]
            evals = j["config"]["pixels"] * j["config"]["components"] * j["roofline"]["walker_steps_per_launch"]
            print(f"tier {t} {name:7s}: {ms:8.3f} ms/launch  {evals / ms / 1e6:8.1f} Gevals/s  {ms * 1e-3 * 39.3e12 / evals:6.1f} dfma-slots/eval", flush=True)
        except Exception as e:
            print("tier", t, "failed", e, out.stderr[-500:])
