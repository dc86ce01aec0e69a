#!/usr/bin/env python3
"""Time the PRODUCTION kernel on workloads whose every evaluation falls in one branch of the Voigt
evaluator (lines placed outside the region through custom centroid bounds), to compare with the
timing-only forced-branch builds of tools/variants.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vamp_amd

P, K, W = 16384, 16, 65536
x = np.arange(P, dtype=np.float64)
rng = np.random.default_rng(0)
flux = 1.0 + rng.normal(0, 0.01, P)
noise = np.full(P, 0.01)
# (name, r at x=0, r at x=P-1): r = 1.665 (x - c)/G
for name, r0, r1 in (("2-level", 110.0, 900.0), ("3-level", 26.0, 98.0), ("4-level", 14.2, 24.5), ("6-level", 8.1, 13.8),
                     ("near-axis", 0.5, 7.5)):
    G = 1.6651092223153954 * (P - 1) / (r1 - r0)
    c = -r0 * G / 1.6651092223153954
    th = np.empty((W, 4 * K))
    for k in range(K):
        th[:, 4 * k] = 1e-3 * (1 + 1e-3 * rng.standard_normal(W))
        th[:, 4 * k + 1] = c * (1 + 1e-6 * rng.standard_normal(W)) - 1e-3 * k
        th[:, 4 * k + 2] = 0.1 * G * (1 + 1e-3 * rng.standard_normal(W))
        th[:, 4 * k + 3] = G * (1 + 1e-4 * rng.standard_normal(W))
    ctx = vamp_amd.HipContext(0)
    ctx.set_regions(x, flux, noise, K, mode=vamp_amd.MODE_VOIGT4, bounds=np.array([[-1e9, 1e9, 1e9, 1e9]]))
    ctx.sampler_init(th, seed=1)
    ctx.run(1, store_chain=False)
    ctx.kernel_timing(True)
    ctx.run(2, store_chain=False)
    ms, n = ctx.kernel_timing(False)
    ms /= n
    evals = P * K * (W // 2)
    print(f"{name:10s} {ms:8.3f} ms/launch  {ms * 1e-3 * 39.3e12 / evals:6.1f} slots/eval", flush=True)
    ctx.close()
