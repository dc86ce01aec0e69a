#!/usr/bin/env python3
"""BASELINE.json config 2: H I Ly-alpha region [672,716] of simba_H1215 (44 px), 4 Voigt components,
4096 walkers, fp64, one GPU.   python tools/bench_c2.py [--steps 2000] [--walkers 4096] [--comp 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=4096)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--comp", type=int, default=4)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--packing", type=int, default=0)
ap.add_argument("--region", type=int, default=0, help="which H I region of the simba spectrum (0: 44 px)")
ap.add_argument("--regions", type=int, default=1, help="copies of the region in one context (a model-selection rung: hundreds of regions x tens of walkers)")
ap.add_argument("--resident", type=int, default=1, help="0: one launch per half-step; 1: the device-resident loop where the library's policy takes it; 2: wherever it can run")
a = ap.parse_args()
import vamp_amd
from vamp_amd.physics import Wave2freq
g = np.load(os.path.join(ROOT, "tests", "golden", "simba_spectra.npz"))
s, e = g["H1215_region_pixels"][a.region]
nu = np.flip(Wave2freq(g["H1215_wavelength"][s:e]), 0)
flux, noise = np.flip(g["H1215_flux"][s:e], 0), np.flip(g["H1215_noise"][s:e], 0)
x = (nu - 0.5 * (nu[0] + nu[-1])) / ((nu[-1] - nu[0]) / (nu.size - 1))
rng = np.random.default_rng(2)
K, W = a.comp, a.walkers
th = np.empty((W, 4 * K))
for k in range(K):
    th[:, 4 * k] = rng.gamma(2.0, 1.0, W)
    th[:, 4 * k + 1] = rng.uniform(x[0], x[-1], W)
    th[:, 4 * k + 2] = rng.uniform(0.5, 8, W)
    th[:, 4 * k + 3] = rng.uniform(2, 15, W)
ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F64 if a.dtype == "f64" else vamp_amd.F32)
ctx.set_packing(a.packing)
ctx.set_option("resident", a.resident)
if a.regions > 1:
    ctx.set_regions([x] * a.regions, [flux] * a.regions, [noise] * a.regions, [K] * a.regions, mode=vamp_amd.MODE_VOIGT4)
    ctx.sampler_init([th] * a.regions, seed=5)
else:
    ctx.set_regions(x, flux, noise, K, mode=vamp_amd.MODE_VOIGT4)
    ctx.sampler_init(th, seed=5)
ctx.run(50, store_chain=False)
t0 = time.perf_counter()
res = ctx.run(a.steps, thin=10, store_chain=a.regions == 1)     # wall clock without per-launch events
dt = time.perf_counter() - t0
ctx.kernel_timing(True)
ctx.run(a.steps, thin=10, store_chain=a.regions == 1)           # same again with HIP events around every launch
ms, n = ctx.kernel_timing(False)
nacc = np.concatenate([np.ravel(v) for v in res["n_accept"]]) if a.regions > 1 else res["n_accept"]
print(json.dumps({"config": f"simba H I region, P={x.size}, K={K}, W={W}, {a.dtype}" + (f", {a.regions} regions" if a.regions > 1 else ""), "walker_steps_per_s": W * a.regions * a.steps / dt,
                  "us_per_half_step_wall": dt / a.steps / 2 * 1e6,
                  # HIP events: per launch on the launch-per-half-step path; the device-resident loop is ONE launch for the whole run
                  "us_per_half_step_kernel": (ms / max(1, n) * 1e3) if n >= 2 * a.steps else ms * 1e3 / (2 * a.steps),
                  "faddeeva_gevals_per_s": W * a.steps * x.size * K / dt / 1e9,
                  "acceptance_fraction": float(nacc.mean()) / (a.steps + 50), "packing": a.packing,
                  "resident_requested": bool(a.resident), "launches_timed": n}))
