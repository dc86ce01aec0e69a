#!/usr/bin/env python3
"""Developer soak (not collected by pytest): stretch-move trajectories of random multi-region contexts -- every
parameterisation, every packing that serves short regions, several split-block sizes -- against the oracle's sampler
driven by the same counter-based draws: identical accept / reject decisions, positions to 1e-10.
usage (GPU box): python tests/soak_sampler.py [n_contexts]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vamp_amd                                   # noqa: E402
from oracle import vamp_oracle as vo              # noqa: E402
from soak_short_regions import make               # noqa: E402  (its __main__ part is guarded below)

if __name__ == "__main__":
    n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    bad = 0
    for packing in (0, 16, 64, 65):
        for c in range(n_ctx):
            rng = np.random.default_rng(12000 + c)
            variant = c % 4
            W = int(rng.choice([32, 64, 96]))
            block = int(rng.choice([b for b in (8, 16, 32) if W % b == 0]))
            xs, fs, ns, Ks, ths, regs, nbz, kw = make(rng, 4, W, variant)
            for t in ths:                              # walkers inside the priors: the soak of the log-posterior covers the rest
                t[W // 8: W // 4, 0] = np.abs(t[W // 8: W // 4, 0]) + 0.05
            ctx = vamp_amd.HipContext(device=0)
            ctx.set_packing(packing)
            ctx.set_regions(xs, fs, ns, Ks, nbz=nbz, **kw)
            seed = 100 + c
            ctx.sampler_init(ths, seed=seed, split_block=block)
            res = ctx.run(3)
            ctx.close()
            for r in range(4):
                fn = lambda q, r=r: vo.log_prob_batch_fast(regs[r], q)
                chain, lch, nacc = vo.run_sampler(fn, ths[r], fn(ths[r]), 3, seed=seed, block=block, region=r, walker_off=r * W)
                ok = np.array_equal(res["n_accept"][r], nacc) and np.allclose(res["chain"][r], chain, rtol=1e-10, atol=1e-12)
                if not ok:
                    bad += 1
                    print("FAIL", packing, c, variant, r, W, block, len(xs[r]), Ks[r], int(np.sum(res["n_accept"][r] != nacc)), flush=True)
        print(f"packing {packing}: {n_ctx} contexts x 4 regions x 3 steps", flush=True)
    assert bad == 0
    print("soak ok")
