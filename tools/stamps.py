#!/usr/bin/env python3
"""Where a latency-bound wavefront's time goes: in-kernel shader-clock stamps of ONE walker's path through a half-step
(workgroup 0, lane 0), from a -DVAMP_STAMPS build of the library.

    python tools/variants.py build stamps=-DVAMP_STAMPS                (here; cross-compiles)
    python tools/stamps.py [--walkers 32] [--comp 1] [--resident 1]     (GPU box)

tags: 0 half-step start, 1 draws in hand, 2 proposal in LDS (the two rows and lnprob have arrived), 3 staged (line records,
prior, near-axis tables), 4 pixels swept and reduced, 5 log-posterior, 6 accept + state written, 7 before / 8 after the
half-step barrier (resident loop only)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=32)
ap.add_argument("--comp", type=int, default=1)
ap.add_argument("--resident", type=int, default=1)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--regions", type=int, default=1, help="copies of the region in the context (workgroup 0 is stamped)")
ap.add_argument("--packing", type=int, default=0, help="force a launch shape (16: four walkers per wavefront; 64: one)")
a = ap.parse_args()
os.environ["VAMP_HIP_LIB"] = os.path.join(ROOT, "build", "variants", "lib_stamps.so")
import vamp_amd                                                        # noqa: E402
from vamp_amd import _lib                                              # noqa: E402
from vamp_amd.physics import Wave2freq                                 # noqa: E402
g = np.load(os.path.join(ROOT, "tests", "golden", "simba_spectra.npz"))
s, e = g["H1215_region_pixels"][0]
nu = np.flip(Wave2freq(g["H1215_wavelength"][s:e]), 0)
flux, noise = np.flip(g["H1215_flux"][s:e], 0), np.flip(g["H1215_noise"][s:e], 0)
x = (nu - 0.5 * (nu[0] + nu[-1])) / ((nu[-1] - nu[0]) / (nu.size - 1))
rng = np.random.default_rng(2)
K, W, R = a.comp, a.walkers, a.regions
th = np.empty((W, 4 * K))
for k in range(K):
    th[:, 4 * k] = rng.gamma(2.0, 1.0, W)
    th[:, 4 * k + 1] = rng.uniform(x[0], x[-1], W)
    th[:, 4 * k + 2] = rng.uniform(0.5, 8, W)
    th[:, 4 * k + 3] = rng.uniform(2, 15, W)
ctx = vamp_amd.HipContext(device=0, dtype=a.dtype)
ctx.set_option("resident", a.resident)
ctx.set_packing(a.packing)
ctx.set_regions([x] * R, [flux] * R, [noise] * R, [K] * R, mode=vamp_amd.MODE_VOIGT4)
ctx.sampler_init([th] * R, seed=5)
ctx.run(50, store_chain=False)
lib = _lib.load()
buf = (C.c_ulonglong * (2 * 8192))()
lib.vamp_debug_stamps.restype = C.c_int
lib.vamp_debug_stamps(buf, 8192)                                        # reset
ctx.run(a.steps, store_chain=False)
n = lib.vamp_debug_stamps(buf, 8192)
st = np.array(buf[:2 * n], dtype=np.uint64).reshape(n, 2)
tags, clk = st[:, 0].astype(int), st[:, 1].astype(np.int64)
# split into half-steps at tag 0; average the intervals between consecutive stamps by (from, to) tag pair
sums, counts = {}, {}
starts = np.nonzero(tags == 0)[0]
for i0, i1 in zip(starts[:-1], starts[1:]):
    for j in range(i0, i1 - 1):
        key = (int(tags[j]), int(tags[j + 1]))
        sums[key] = sums.get(key, 0) + int(clk[j + 1] - clk[j])
        counts[key] = counts.get(key, 0) + 1
    key = (int(tags[i1 - 1]), 0)
    sums[key] = sums.get(key, 0) + int(clk[i1] - clk[i1 - 1])
    counts[key] = counts.get(key, 0) + 1
half = np.diff(clk[starts]).mean() if starts.size > 1 else float("nan")
out = {"config": f"P={x.size} K={K} W={W} regions={R} resident={a.resident} {a.dtype} packing={a.packing}", "stamps": int(n), "half_steps": int(starts.size),
       "clocks_per_half_step_same_workgroup": float(half),
       "mean_clocks": {f"{k[0]}->{k[1]}": round(sums[k] / counts[k], 1) for k in sorted(sums)},
       "counts": {f"{k[0]}->{k[1]}": counts[k] for k in sorted(counts)}}
print(json.dumps(out))
