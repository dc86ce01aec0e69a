#!/usr/bin/env python3
"""Developer soak (not collected by pytest): the two device-resident loops of round 4 against their launch-per-step
forms on RANDOM contexts -- every parameterisation, packing, ragged region mixes (1..8 lines, 9..330 px, now and then a
region of 17..24 lines), even walker counts that fill their wavefronts or do not, thinning, continued runs:
  * k_run_resident ("resident" = 2) vs one launch per half-step ("resident" = 0): chain, log-posterior chain,
    acceptance counts and final state bit for bit;
  * k_map_search ("map_device" = 1) vs the host-driven search (0): optimum bit for bit, same iteration counts.
usage (GPU box): python tests/soak_resident_map.py [n_contexts]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vamp_amd                                   # noqa: E402

C_LIGHT, SIGMA0, LINE, PIX_HZ = 2.98e8, 0.0263, 1215.67, 4.0e10
FPS = 2.0 * np.sqrt(2.0 * np.log(2.0))
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def make(rng, variant, W, packing):
    n_regions = int(rng.integers(1, 9))
    xs, fs, ns, Ks, ths, nbz = [], [], [], [], [], []
    for _ in range(n_regions):
        P = int(rng.choice([9, 14, 23, 36, 51, 64, 65, 97, 130, 200, 257, 330]))
        K = int(rng.integers(1, 9))
        if packing in (0, 64) and rng.random() < 0.1:
            K = int(rng.integers(17, 25))
        x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
        c = rng.uniform(x[0] * 0.8, x[-1] * 0.8, K)
        w = rng.uniform(1.0, 0.06 * P + 2.0, K)
        tau = sum(rng.uniform(0.3, 2.0) * np.exp(-0.5 * ((x - ck) / wk) ** 2) for ck, wk in zip(c, w))
        th = np.empty((W, K, 4))
        th[:, :, 0] = rng.uniform(0.2, 1.5, (W, K))
        th[:, :, 1] = c + rng.normal(0, 1.5, (W, K))
        th[:, :, 2] = 10.0 ** rng.uniform(-2, 0.5, (W, K))
        th[:, :, 3] = FPS * w * rng.uniform(0.6, 1.6, (W, K))
        th[: max(1, W // 8), 0, 0] = -0.1
        xs.append(x); fs.append(np.exp(-tau) + rng.normal(0, 0.02, P)); Ks.append(K)
        ns.append(np.ones(P) if variant == 2 else np.full(P, 0.02))
        if variant == 1:
            t = np.stack([th[:, :, 0], th[:, :, 1], th[:, :, 3] / FPS], axis=2).reshape(W, 3 * K)
        elif variant == 2:
            t = np.hstack([th.reshape(W, 4 * K), rng.uniform(0.01, 0.2, (W, 1))])
        elif variant == 3:
            nu_mid = C_LIGHT / (1225.0 * 1e-10)
            sig_hz = th[:, :, 3] * PIX_HZ / FPS
            t = np.stack([th[:, :, 0] * sig_hz * np.sqrt(2 * np.pi) / SIGMA0, (LINE * 1e-10 * sig_hz * 2.355 / np.sqrt(2)) * 1e-3,
                          ((C_LIGHT / (nu_mid + PIX_HZ * th[:, :, 1])) / 1e-10 - LINE) / LINE], axis=2).reshape(W, 3 * K)
            nbz.append([float(10.0 ** rng.uniform(-1, 0.5)), LINE, nu_mid, PIX_HZ])
        else:
            t = th.reshape(W, 4 * K)
        ths.append(np.ascontiguousarray(t))
    kw = [dict(mode=vamp_amd.MODE_VOIGT4), dict(mode=vamp_amd.MODE_GAUSS3), dict(mode=vamp_amd.MODE_VOIGT4, sample_sd=True),
          dict(mode=vamp_amd.MODE_NBZ3)][variant]
    if variant == 3:
        kw["nbz"] = np.array(nbz)
    return xs, fs, ns, Ks, ths, kw


bad = 0
for c in range(n_ctx):
    rng = np.random.default_rng(31000 + c)
    variant, packing = c % 4, [0, 16, 64, 65, 0][(c // 4) % 5]
    W = int(2 * rng.integers(4, 100))
    divs = [b for b in range(2, W + 1, 2) if W % b == 0]
    block = int(rng.choice(divs))
    xs, fs, ns, Ks, ths, kw = make(rng, variant, W, packing)
    if packing in (16, 65) and max(Ks) > 8:
        continue
    ctx = vamp_amd.HipContext(device=0)
    ctx.set_packing(packing)
    out = {}
    n1, t1, n2 = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
    for resident in (2, 0):
        ctx.set_option("resident", resident)
        ctx.set_regions(xs, fs, ns, Ks, **kw)
        ctx.sampler_init(ths, seed=1000 + c, split_block=block)
        a = ctx.run_flat(n1, thin=t1)
        b = ctx.run_flat(n2, thin=1)
        out[resident] = (a, b, ctx.get_state())
    (a1, b1, s1), (a0, b0, s0) = out[2], out[0]
    ok = all(np.array_equal(u, v) for x1, x0 in ((a1, a0), (b1, b0)) for u, v in zip(x1[:3], x0[:3]))
    flat = lambda st: [np.concatenate([np.ravel(v) for v in (st[i] if isinstance(st[i], list) else [st[i]])]) for i in range(3)]
    ok = ok and all(np.array_equal(u, v) for u, v in zip(flat(s1), flat(s0))) and s1[3] == s0[3]
    # the MAP searches from every region's first finite walker
    starts = [t[np.argmax(np.isfinite(ctx.lnprob(t, region=r)))] for r, t in enumerate(ths)]
    kwm = dict(iterlim=int(rng.integers(5, 120)), tol=float(10.0 ** rng.uniform(-6, -2)), xtol=float(10.0 ** rng.uniform(-6, -2)))
    ctx.set_option("map_device", 1)
    m1 = ctx.map_all(starts, **kwm)
    ctx.set_option("map_device", 0)
    m0 = ctx.map_all(starts, **kwm)
    okm = np.array_equal(m1[3], m0[3]) and all(np.array_equal(u, v) for u, v in zip(m1[0], m0[0])) and np.array_equal(m1[1], m0[1])
    ctx.close()
    if not (ok and okm):
        bad += 1
        print("FAIL context", c, "variant", variant, "packing", packing, "W", W, "block", block, "regions", list(zip([len(x) for x in xs], Ks)),
              "sampler" if not ok else "", "map" if not okm else "", flush=True)
print(f"{n_ctx} contexts, {bad} failures")
assert bad == 0
print("soak ok")
