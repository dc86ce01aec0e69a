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


C_LIGHT, SIGMA0, LINE, PIX_HZ = 2.98e8, 0.0263, 1215.67, 4.0e10          # physics.py:3-4 and a simba-like pixel
FWHM_PER_SIGMA = 2.0 * np.sqrt(2.0 * np.log(2.0))


def make(rng, n_regions, W, variant, kmin=1, kmax=8):
    """variant 0: (amplitude, centroid, L, G); 1: Gaussian components; 2: variant 0 with the reference's free
    precision sd as last dimension (vpfits.py:39); 3: (N, b, z) through the reference's maps"""
    xs, fs, ns, Ks, ths, regs, nbzs = [], [], [], [], [], [], []
    for _ in range(n_regions):
        P = int(rng.choice([9, 14, 23, 36, 51, 64, 65, 97, 130, 200, 256, 257, 330, 478, 512]))
        K = int(rng.integers(kmin, kmax + 1))
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
        noise = np.full(P, 0.03) if variant != 2 else np.ones(P)
        flux = np.clip(1.0 + rng.normal(0, 0.03, P), 0, None)
        kw = {}
        if variant == 1:
            t = np.stack([th[:, :, 0], th[:, :, 1], th[:, :, 3] / FWHM_PER_SIGMA], axis=2).reshape(W, 3 * K)
            kw = dict(mode=vo.MODE_GAUSS3)
        elif variant == 2:
            t = np.hstack([th.reshape(W, 4 * K), rng.uniform(0.005, 0.3, (W, 1))])
            kw = dict(mode=vo.MODE_VOIGT4, sample_sd=True)
        elif variant == 3:
            l_fixed = float(10.0 ** rng.uniform(-2, 1))
            nu_mid = C_LIGHT / (1225.0 * 1e-10)
            sig_hz = th[:, :, 3] * PIX_HZ / FWHM_PER_SIGMA
            Ncol = th[:, :, 0] * sig_hz * np.sqrt(2 * np.pi) / SIGMA0
            b = (LINE * 1e-10 * sig_hz * 2.355 / np.sqrt(2)) * 1e-3
            zred = ((C_LIGHT / (nu_mid + PIX_HZ * th[:, :, 1])) / 1e-10 - LINE) / LINE
            t = np.stack([Ncol, b, zred], axis=2).reshape(W, 3 * K)
            kw = dict(mode=vo.MODE_NBZ3)
            nbzs.append([l_fixed, LINE, nu_mid, PIX_HZ])
        else:
            t = th.reshape(W, 4 * K)
            kw = dict(mode=vo.MODE_VOIGT4)
        xs.append(x); fs.append(flux); ns.append(noise); Ks.append(K); ths.append(np.ascontiguousarray(t))
        r = vo.Region(x=x, flux=flux, noise=noise, n_comp=K, **kw)
        if variant == 3:
            r.l_fixed, r.line, r.x_origin, r.x_scale = nbzs[-1]
        regs.append(r)
    return xs, fs, ns, Ks, ths, regs, (np.array(nbzs) if variant == 3 else None), kw


if __name__ == "__main__":
    worst = {}
    # (packing, walkers, lines per region): the last two rows are regions of 17 .. 32 lines (their own launch class)
    for packing, W, (kmin, kmax) in ((0, 64, (1, 8)), (16, 64, (1, 8)), (64, 64, (1, 8)), (65, 64, (1, 8)), (0, 16384, (1, 8)),
                                     (0, 280, (9, 32)), (64, 280, (17, 32))):
        w = 0.0
        for c in range(n_ctx if W == 64 else max(1, n_ctx // 10)):
            rng = np.random.default_rng(9000 + c)
            variant = c % 4
            if F32 and variant == 1:
                variant = 0                              # (Gaussian components have no W4 path to test)
            xs, fs, ns, Ks, ths, regs, nbz, kw = make(rng, 8 if W == 64 else 3, W, variant, kmin, kmax)
            ctx = vamp_amd.HipContext(device=0, dtype=vamp_amd.F32 if F32 else vamp_amd.F64)
            ctx.set_packing(packing)
            ctx.set_regions(xs, fs, ns, Ks, nbz=nbz, **kw)
            got, chi = ctx.lnprob_all(ths, return_chi2=True)
            ctx.close()
            if F32:
                # SURVEY 8d states the fp32 tolerance on chi^2 (with a free precision sd the log-posterior is a difference of
                # two large terms and its relative error is not the sum's): chi^2 of the fp64 device path -- itself compared
                # with the oracle by the fp64 run of this soak -- is the reference
                ctx = vamp_amd.HipContext(device=0)
                ctx.set_packing(packing)
                ctx.set_regions(xs, fs, ns, Ks, nbz=nbz, **kw)
                ref, chi_ref = ctx.lnprob_all(ths, return_chi2=True)
                ctx.close()
            for r in range(len(xs)):
                if F32:
                    if not np.array_equal(np.isfinite(ref[r]), np.isfinite(got[r])):
                        print("FAIL pattern", packing, W, c, variant, r, len(xs[r]), Ks[r], flush=True)
                        w = 1.0
                        continue
                    fin = np.isfinite(ref[r])
                    if fin.any():
                        err = np.abs(chi[r][fin] - chi_ref[r][fin]) / np.maximum(chi_ref[r][fin], 1e-300)
                        w = max(w, err.max())
                        if err.max() > TOL:
                            print("FAIL", packing, W, c, variant, r, len(xs[r]), Ks[r], err.max(), flush=True)
                    continue
                want = vo.log_prob_batch_fast(regs[r], ths[r])
                if not np.array_equal(np.isfinite(want), np.isfinite(got[r])):
                    print("FAIL pattern", packing, W, c, variant, r, len(xs[r]), Ks[r], flush=True)
                    w = 1.0
                    continue
                fin = np.isfinite(want)
                if fin.any():
                    err = np.abs(got[r][fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
                    w = max(w, err.max())
                    if err.max() > TOL:
                        print("FAIL", packing, W, c, variant, r, len(xs[r]), Ks[r], err.max(), flush=True)
        worst[(packing, W, kmax)] = w
        print(f"packing {packing}, {W} walkers, {kmin}..{kmax} lines: worst relative {'chi^2' if F32 else 'lnprob'} error {w:.3e}", flush=True)
    assert max(worst.values()) <= TOL
    print("soak ok")
