#!/usr/bin/env python3
"""BASELINE.json config 5, "HBM-BW roofline sweep": the q1422 workload of config 3 (421 regions batched)
in fp32 / Humlicek W4 and in fp64, over the ensemble size -- region-walker-steps/s, algorithmic GB/s
(SURVEY 8d bytes) and the fraction of the 8 TB/s HBM peak it corresponds to.

    python tools/sweep_c5.py [steps]      (GPU box; one tools/bench_c3.py process per point)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
print(f"{'dtype':5s} {'walkers':>8s} {'ms/half-step':>13s} {'M region-walker-steps/s':>24s} {'alg GB/s':>9s} {'HBM frac':>9s} {'G Re w / s (nominal)':>21s}")
for dtype in ("f32", "f64"):
    for W in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_c3.py"), "--steps", steps, "--walkers", str(W),
                              "--dtype", dtype], capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:          # noqa: BLE001
            print(dtype, W, "failed:", e, out.stderr[-300:], flush=True)
            continue
        print(f"{dtype:5s} {W:8d} {j['avg_launch_ms']:13.3f} {j['region_walker_steps_per_s'] / 1e6:24.1f} "
              f"{j['algorithmic_GBps']:9.1f} {j['hbm_frac']:9.4f} {j['faddeeva_gevals_per_s']:21.1f}", flush=True)
