#!/usr/bin/env python3
"""BASELINE.json config 3: every detected region of the q1422 spectrum (421 regions, 9..478 px)
sampled in ONE batched launch per half-step, <= 8 Voigt components each.

    python tools/bench_c3.py [--walkers 16384] [--steps 5] [--dtype f64|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_c3.py ...
        (N GPUs: the regions are spread over the ranks by W * P * K, no collective -- vamp_amd.ensemble.RegionShardedBatch)

Regions and the spectrum come from tests/golden/q1422_spectrum.npz (the reference's
vamp_1.0/data/q1422.cont and the detector of vpspectrum.py:67-175)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_regions(max_comp=8):
    from vamp_amd.physics import Wave2freq
    from vamp_amd.vpregion import VPregion
    q = np.load(os.path.join(ROOT, "tests", "golden", "q1422_spectrum.npz"))
    wl, fl, no = q["wavelength_milli"] / 1000.0, q["flux_micro"] / 1e6, q["noise_micro"] / 1e6
    nu = Wave2freq(wl)
    xs, fs, ns, ks = [], [], [], []
    for s, e in q["region_pixels"]:
        f = np.flip(nu[s:e], 0)
        mid, d = 0.5 * (f[0] + f[-1]), (f[-1] - f[0]) / (f.size - 1)
        flux, noise = np.flip(fl[s:e], 0), np.flip(no[s:e], 0)
        xs.append((f - mid) / d)
        fs.append(flux)
        ns.append(noise)
        ks.append(min(max_comp, VPregion(f, flux, noise).n))
    return xs, fs, ns, ks


def start_walkers(rng, x, K, W):
    span = x[-1] - x[0]
    th = np.empty((W, 4 * K))
    for k in range(K):
        th[:, 4 * k + 0] = rng.gamma(2.0, 0.5, W)
        th[:, 4 * k + 1] = rng.uniform(x[0], x[-1], W)
        th[:, 4 * k + 2] = rng.uniform(0.02, 0.3, W) * span
        th[:, 4 * k + 3] = rng.uniform(0.05, 0.5, W) * span
    return th


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--packing", type=int, default=0, help="lanes per walker: 0 auto, 16, 64")
    a = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these boxes
    import vamp_amd
    xs, fs, ns, ks = build_regions()
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    rng = np.random.default_rng(1422)
    theta0 = [start_walkers(rng, x, k, a.walkers) for x, k in zip(xs, ks)]
    dist = None
    if world > 1:
        import torch.distributed as dist          # host-side bootstrap / barrier only (gloo)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctx = vamp_amd.HipContext(device=local, dtype=vamp_amd.F64 if a.dtype == "f64" else vamp_amd.F32)
    ctx.set_packing(a.packing)
    from vamp_amd.ensemble import RegionShardedBatch
    batch = RegionShardedBatch(ctx, xs, fs, ns, ks, theta0, seed=1422, mode=vamp_amd.MODE_VOIGT4, dist=dist,
                               split_block=vamp_amd.default_split_block(a.walkers))
    batch.run(a.warmup, store_chain=False)
    if dist is not None:
        dist.barrier()
    ctx.kernel_timing(True)
    t0 = time.perf_counter()
    mine, _ = batch.run(a.steps, store_chain=False)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    ms, n = ctx.kernel_timing(False)
    acc_mine = [float(v["n_accept"].mean()) for v in mine.values()]
    if dist is not None:
        import torch
        t = torch.tensor([dt])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        box = [None] * world
        dist.all_gather_object(box, acc_mine)
        acc_mine = sum(box, [])
    if rank != 0:
        return
    R = len(xs)
    P = np.array([len(x) for x in xs])
    K = np.array(ks)
    D = 4 * K
    s = 8 if a.dtype == "f64" else 4
    b_alg = float(np.sum(a.walkers * (3 * P + 3 * D + 2) * s))          # per step, all regions
    evals = float(np.sum(a.walkers * P * K))
    acc = np.mean(acc_mine) / (a.steps + a.warmup)
    print(json.dumps({"config": "q1422: %d regions, sum P = %d, sum K = %d, W = %d per region, %s" % (R, P.sum(), K.sum(), a.walkers, a.dtype),
                      "n_gpus": world, "regions_on_rank0": len(batch.mine),
                      "region_walker_steps_per_s": R * a.walkers * a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
                      "avg_launch_ms": ms / max(1, n), "faddeeva_gevals_per_s": evals * a.steps / dt / 1e9,
                      "algorithmic_GBps": b_alg * a.steps / dt / 1e9, "hbm_frac": b_alg * a.steps / dt / 8e12 / world,
                      "acceptance_fraction": float(acc), "packing": a.packing}))


if __name__ == "__main__":
    main()
