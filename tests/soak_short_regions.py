#!/usr/bin/env python3
"""Developer soak (not collected by pytest): random SHORT regions -- the shapes of real spectra: 9..500 pixels, 1..8
lines, several regions per context so that the launch classes (blends, single-line regions, the rest) are all
populated -- with line widths and dampings spread over decades, under every packing that serves them, against the
oracle.  usage (GPU box): python tests/soak_short_regions.py [n_contexts] [f32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vamp_amd                                   # noqa: E402
from oracle import vamp_oracle as vo              # noqa: E402

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
TOL = 1e-3 if F32 else 1e-9


def make(rng, n_regions, W):
    xs, fs, ns, Ks, ths, regs = [], [], [], [], [], []
    for _ in range(n_regions):
        P = int(rng.choice([9, 14, 23, 36, 51, 64, 65, 97, 130, 200, 256, 257, 330, 478, 512]))
        K = int(rng.integers(1, 9))
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        if rng.random() < 0.3:
            x = np.cumsum(rng.uniform(0.7, 1.3, P))
            x -= x.mean()
        span = x[-1] - x[0]
        th = np.empty((W, K, 4))
        th[:, :, 0] = 10.0 ** rng.uniform(-2, 1.5, (W, K))
        th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))
        th[:, :, 2] = 10.0 ** rng.uniform(-6, np.log10(0.9 * span), (W, K))
        th[:, :, 3] = 10.0 ** rng.uniform(-1.3, np.log10(0.9 * span), (W, K))
        th[: W // 8, 0, 1] = rng.choice([x[0], x[-1]], W // 8)          # some centres on an edge
        th[W // 8: W // 4, 0, 0] = -0.1                                  # some outside the prior
        noise = np.full(P, 0.03)
        flux = np.clip(1.0 + rng.normal(0, 0.03, P), 0, None)
        xs.append(x); fs.append(flux); ns.append(noise); Ks.append(K); ths.append(th.reshape(W, 4 * K))
        regs.append(vo.Region(x=x, flux=flux, noise=noise, n_comp=K, mode=vo.MODE_VOIGT4))
    return xs, fs, ns, Ks, ths, regs


worst = {}
for packing, W in ((0, 64), (16, 64), (64, 64), (65, 64), (0, 16384)):
    w = 0.0
    for c in range(n_ctx if W == 64 else max(1, n_ctx // 10)):
        rng = np.random.default_rng(9000 + c)
        xs, fs, ns, Ks, ths, regs = make(rng, 8 if W == 64 else 3, W)
        ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F32 if F32 else vamp_amd.F64)
        ctx.set_packing(packing)
        ctx.set_regions(xs, fs, ns, Ks, mode=vo.MODE_VOIGT4)
        got = ctx.lnprob_all(ths)
        ctx.close()
        for r in range(len(xs)):
            want = vo.log_prob_batch_fast(regs[r], ths[r])
            if not np.array_equal(np.isfinite(want), np.isfinite(got[r])):
                print("FAIL pattern", packing, W, c, r, len(xs[r]), Ks[r], flush=True)
                w = 1.0
                continue
            fin = np.isfinite(want)
            if fin.any():
                err = np.abs(got[r][fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
                w = max(w, err.max())
                if err.max() > TOL:
                    print("FAIL", packing, W, c, r, len(xs[r]), Ks[r], err.max(), flush=True)
    worst[(packing, W)] = w
    print(f"packing {packing}, {W} walkers: worst relative lnprob error {w:.3e}", flush=True)
assert max(worst.values()) <= TOL
print("soak ok")
