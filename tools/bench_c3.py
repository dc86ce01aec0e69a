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
    # roofline object in the form of bench.py's (contract: algorithmic bytes against the HBM peak; the binding roof is
    # VALU issue, quoted from the committed PMC passes of this command -- profiles/pmc_traffic_c3*.json)
    from bench import committed_pmc, VALU_PEAK_LANE_INSTR, VALU_PEAK_LANE_INSTR_F32
    half_ms = ms / max(1, n)                       # HIP events around ALL launches of a half-step (the classes run concurrently)
    pj = committed_pmc(0, 0, a.walkers, 0, world, a.dtype, workload="q1422") if world == 1 and a.packing == 0 else None
    valu_peak = VALU_PEAK_LANE_INSTR if a.dtype == "f64" else VALU_PEAK_LANE_INSTR_F32
    roof = {"bound": "hbm", "achieved": b_alg / 2 / (half_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": b_alg / 2 / (half_ms * 1e-3) / 8e12,
            "achieved_is": "algorithmic GB/s: sum over regions of W (3 P + 3 D + 2) s bytes per step (SURVEY 8d), half of it per half-step, "
                           "/ the HIP-event time of a half-step (all launch classes)",
            "traffic": None if pj is None else (2.0 * pj["FETCH_SIZE_KB_per_launch"] + pj["WRITE_SIZE_KB_per_launch"]) * 1024.0,
            "traffic_is": "HBM bytes per half-step, (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB summed over the half-step's kernels, from the committed PMC passes",
            "traffic_source": None if pj is None else pj["source"],
            "kernel": "k_half_step (one launch per launch class) + k_draws", "avg_half_step_ms": half_ms, "half_steps": n,
            "algorithmic_bytes_per_half_step": b_alg / 2,
            "binding_roof": "%s VALU issue" % ("fp64" if a.dtype == "f64" else "fp32"),
            "valu": None if pj is None else {
                "peak_lane_instr_per_s": valu_peak,
                "executed_lane_instr_per_s": pj["SQ_INSTS_VALU_per_launch"] * 64 / (half_ms * 1e-3),
                "valu_issue_frac": pj["SQ_INSTS_VALU_per_launch"] * 64 / (half_ms * 1e-3) / valu_peak,
                "valu_busy_frac_pmc": pj.get("valu_busy_frac_pmc"),
                "formula": "sum over the half-step's kernels of SQ_INSTS_VALU (wave instructions, committed PMC pass) x 64 lanes / "
                           "half-step time / (256 CUs x 4 SIMDs x %d lanes/clk x 2.4 GHz)" % (16 if a.dtype == "f64" else 32)}}
    print(json.dumps({"config": "q1422: %d regions, sum P = %d, sum K = %d, W = %d per region, %s" % (R, P.sum(), K.sum(), a.walkers, a.dtype),
                      "metric": "region-walker-steps/sec", "dtype": a.dtype,
                      "n_gpus": world, "regions_on_rank0": len(batch.mine),
                      "region_walker_steps_per_s": R * a.walkers * a.steps / dt, "ms_per_step": dt / a.steps * 1e3,
                      "avg_launch_ms": ms / max(1, n), "faddeeva_gevals_per_s": evals * a.steps / dt / 1e9,
                      "algorithmic_GBps": b_alg * a.steps / dt / 1e9, "hbm_frac": b_alg * a.steps / dt / 8e12 / world,
                      "roofline": roof,
                      "acceptance_fraction": float(acc), "packing": a.packing}))


if __name__ == "__main__":
    main()
