#!/usr/bin/env python3
"""Headline benchmark: walker-steps/s of the stretch-move + Voigt log-posterior hot path on the
synthetic 16 384-pixel x 16-component x 65 536-walker region of BASELINE.json (config 4).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one full stretch-move step of the whole ensemble (two half-steps: every walker gets
one proposal, one log-posterior evaluation and one accept/reject).  Inputs (spectrum, initial
walker positions) are resident in HBM before the timed region; the chain sample of every step is
recorded device-to-device inside it.  Rank 0 prints ONE JSON line.

Strong scaling: the 65 536 walkers are sharded over the N ranks (contiguous walker blocks), with
one RCCL all-gather of the owned rows per half-step (vamp_amd/ensemble.py).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C_LIGHT = 2.98e8                     # reference constant (physics.py:3)
SIGMA0 = 0.0263
FWHM_PER_SIGMA = 2.0 * np.sqrt(2.0 * np.log(2.0))
SQRT_LN2 = np.sqrt(np.log(2.0))
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak
LINE = 1215.67                       # H I Ly-alpha rest wavelength [A]
PIX_HZ = 4.0e10                      # synthetic pixel width [Hz] (simba H I data: ~4.0e10 Hz)
L_FIXED_PIX = 10.0                   # NBZ3: fixed Lorentzian FWHM in pixels (mean of U(1,20))


def _voigt_tau(x, A, c, L, G):
    from scipy.special import wofz   # workload synthesis only (host, outside any timed region)
    z = (2.0 * (x - c) + 1j * L) * SQRT_LN2 / G
    return A * L * np.sqrt(np.pi) * SQRT_LN2 / G * wofz(z).real


def make_workload(P=16384, K=16, W=65536, seed=20240517, nbz=True):
    """SURVEY section 8d synthetic region.  Region-centred pixel coordinates x_i = i - (P-1)/2;
    truth: centroid ~ U(-0.45P, 0.45P), amplitude ~ U(0.2, 3), G_fwhm ~ U(20, 200) px,
    L_fwhm ~ U(1, 20) px (fixed to 10 px in the 3-parameter NBZ form); sigma = 0.01;
    walkers = truth * (1 + 1e-3 N(0,1)), clipped into the priors of vpfits.py:249-297."""
    rng = np.random.default_rng(seed)
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    scale = P / 16384.0
    c = rng.uniform(-0.45 * P, 0.45 * P, K)
    A = rng.uniform(0.2, 3.0, K)
    G = rng.uniform(20.0, 200.0, K) * max(scale, 1.0 / 16)
    L = rng.uniform(1.0, 20.0, K) * max(scale, 1.0 / 16)
    if nbz:
        L = np.full(K, L_FIXED_PIX * max(scale, 1.0 / 16))
    tau = np.zeros(P)
    for k in range(K):
        tau += _voigt_tau(x, A[k], c[k], L[k], G[k])
    noise = np.full(P, 0.01)
    flux = np.exp(-tau) + rng.normal(0.0, 0.01, P)
    fwhm_max = (x[-1] - x[0]) / 2.0 * FWHM_PER_SIGMA
    out = dict(x=x, flux=flux, noise=noise, K=K, P=P, W=W)
    pert = 1.0 + 1e-3 * rng.standard_normal((W, K, 4))
    nat = np.stack([A, c, L, G], axis=1)[None, :, :] * pert          # (W, K, 4) = A, c, L, G
    nat[:, :, 0] = np.clip(nat[:, :, 0], 1e-6, None)
    nat[:, :, 1] = np.clip(nat[:, :, 1], x[0], x[-1])
    nat[:, :, 2] = np.clip(nat[:, :, 2], 1e-6, fwhm_max)
    nat[:, :, 3] = np.clip(nat[:, :, 3], 1e-6, fwhm_max)
    if not nbz:
        out.update(theta0=np.ascontiguousarray(nat.reshape(W, 4 * K)), mode=1, nbz=None, D=4 * K)
        return out
    # (N, b, z) through the reference's maps (physics.py:15, 27, 120, 134)
    nu_mid = C_LIGHT / (1225.0 * 1e-10)
    sig_hz = nat[:, :, 3] * PIX_HZ / FWHM_PER_SIGMA
    Ncol = nat[:, :, 0] * sig_hz * np.sqrt(2 * np.pi) / SIGMA0
    b = (LINE * 1e-10 * sig_hz * 2.355 / np.sqrt(2)) * 1e-3
    nu_c = nu_mid + PIX_HZ * nat[:, :, 1]
    zred = ((C_LIGHT / nu_c) / 1e-10 - LINE) / LINE
    th = np.stack([Ncol, b, zred], axis=2).reshape(W, 3 * K)
    out.update(theta0=np.ascontiguousarray(th), mode=2, D=3 * K,
               nbz=np.array([[float(L[0]), LINE, nu_mid, PIX_HZ]]))
    return out


def algorithmic_bytes_per_walker_step(P, D, s=8):
    """3 P s (x, flux, 1/sigma read once per evaluation) + 3 D s (own + partner
    position read, new position written) + 2 s (old / new lnprob)."""
    return 3 * P * s + 3 * D * s + 2 * s


def cpu_baseline(wl, budget_s=12.0):
    """C oracle (oracle/vamp_oracle.c, OpenMP over walkers) on the host cores: same region, same
    sampler, a bounded number of walkers for one step.  Test infrastructure used as a reported
    baseline only."""
    so = os.path.join(ROOT, "oracle", "libvamp_oracle.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    lib = C.CDLL(so)
    cores = lib.vo_num_threads()
    dp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None

    def run(Wc, steps):
        X = np.ascontiguousarray(wl["theta0"][:Wc].copy())
        lnp = np.empty(Wc)
        nbz = wl["nbz"]
        lib.vo_lnprob(C.c_int64(wl["P"]), dp(wl["x"]), dp(wl["flux"]), dp(wl["noise"]), wl["K"], wl["mode"], 0, 0, None,
                      dp(nbz), C.c_int64(Wc), dp(X), dp(lnp), None, cores)
        nacc = np.zeros(Wc, dtype=np.int64)
        blk = Wc if Wc <= 1024 else 1024
        t0 = time.perf_counter()
        rc = lib.vo_sampler_run(C.c_int64(wl["P"]), dp(wl["x"]), dp(wl["flux"]), dp(wl["noise"]), wl["K"], wl["mode"], 0, 0,
                                None, dp(nbz), C.c_int64(Wc), dp(X), dp(lnp), dp(nacc), C.c_int64(steps), C.c_int64(0),
                                C.c_uint64(7), C.c_double(2.0), C.c_int32(blk), cores)
        assert rc == 0
        return time.perf_counter() - t0

    w_probe = max(2, 2 * cores)
    t = run(w_probe, 1)
    rate = w_probe / t
    Wc = int(min(wl["W"], max(w_probe, rate * budget_s)))
    Wc -= Wc % 2
    if Wc > 1024:
        Wc -= Wc % 1024
    t = run(Wc, 1)
    return {"value": Wc / t, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "sample": f"C oracle (OpenMP, {cores} threads), first {Wc} walkers of the same workload x 1 step, {t:.1f} s"}


class _StdoutToStderr:
    """RCCL prints a version banner on stdout at communicator creation; keep stdout for the one
    JSON line by pointing fd 1 at stderr while the process group comes up and warms up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--walkers", type=int, default=65536)
    ap.add_argument("--pixels", type=int, default=16384)
    ap.add_argument("--components", type=int, default=16)
    ap.add_argument("--param", choices=["nbz3", "voigt4"], default="nbz3")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="do not record the chain inside the timed region")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the nccl process group and run the all-gather even with one rank (rehearsal)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import vamp_amd

    quiet = _StdoutToStderr()
    quiet.__enter__()
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = make_workload(P=args.pixels, K=args.components, W=args.walkers, nbz=(args.param == "nbz3"))
    P, K, W, D = wl["P"], wl["K"], wl["W"], wl["D"]
    dtype = vamp_amd.F64 if args.dtype == "f64" else vamp_amd.F32
    ctx = vamp_amd.HipContext(device=local_rank, dtype=dtype)
    ctx.set_regions(wl["x"], wl["flux"], wl["noise"], K, mode=wl["mode"], nbz=wl["nbz"])

    from vamp_amd.ensemble import ShardedEnsemble
    torch.cuda.set_device(local_rank)
    # walker state lives in torch tensors on torch's current stream (plumbing only): the RCCL
    # all-gather and the chain record are then ordered with the kernels without host syncs
    ens = ShardedEnsemble(ctx, wl["theta0"], seed=20240517, dist=dist, exchange="nccl" if dist is not None else "none",
                          torch_device=torch.device("cuda", local_rank), torch_state=True, exchange_single_rank=args.force_dist)
    own = ens.own_count

    # chain storage (device resident): each rank records its own rows every step
    chain = None
    if not args.no_chain:
        chain = torch.empty((args.steps, own, D), dtype=torch.float64, device=torch.device("cuda", local_rank))

    def sync_all():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    def record(i):
        if chain is not None:
            off = 0
            for rows in ens._own:                            # D2D on the stream the kernels run on
                chain[i, off:off + rows.shape[0]].copy_(rows, non_blocking=True)
                off += rows.shape[0]

    ens.step(args.warmup)
    sync_all()
    quiet.__exit__()
    ctx.kernel_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ens.step(1)
        record(i)
    sync_all()
    dt = time.perf_counter() - t0
    k_ms, k_n = ctx.kernel_timing(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the ensemble is alive (some proposals accepted, lnprob finite)
    _, lnp, nacc, _ = ctx.get_state()
    acc_frac = float(nacc[ens.own_mask].mean()) / max(1, args.steps + args.warmup)
    finite_frac = float(np.isfinite(lnp[ens.own_mask]).mean())

    if rank == 0:
        value = W * args.steps / dt
        b_alg = algorithmic_bytes_per_walker_step(P, D, 8 if args.dtype == "f64" else 4)
        per_launch_units = own // (2 * ens.parts)       # walker-steps of one half-step launch on this rank
        avg_ms = k_ms / max(1, k_n)
        traffic = None
        try:      # HBM bytes per launch from the committed PMC run of this same command (tools/pmc.sh)
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            c = pj["config"]
            if (c["pixels"], c["components"], c["walkers"], c["ndim"], c["n_gpus"], c["dtype"]) == (P, K, W, D, world, args.dtype):
                traffic = (2.0 * pj["FETCH_SIZE_KB_per_launch"] + pj["WRITE_SIZE_KB_per_launch"]) * 1024.0
        except (OSError, KeyError, ValueError):
            pass
        achieved = per_launch_units * b_alg / (avg_ms * 1e-3) / 1e9 if k_n else None
        line = {
            "metric": "walker-steps/sec (log-posterior evals/sec)",
            "value": value,
            "unit": "walker-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"synthetic region P={P} px x K={K} Voigt components x W={W} walkers, "
                                   f"{'(N,b,z) D=%d' % D if args.param == 'nbz3' else 'native (A,c,L,G) D=%d' % D}, "
                                   f"stretch move a=2, walkers sharded over {world} GPU(s)",
                       "pixels": P, "components": K, "walkers": W, "ndim": D, "parameterisation": args.param,
                       "chain_recorded": chain is not None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes of HBM traffic per launch (PMC, separate passes)",
                         "measured_hbm_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and k_n) else None,
                         "algorithmic_bytes_per_launch": per_launch_units * b_alg,
                         "kernel": "k_half_step", "avg_launch_ms": avg_ms, "launches": k_n,
                         "alg_bytes_per_walker_step": b_alg, "walker_steps_per_launch": per_launch_units},
            "faddeeva_gevals_per_s": value * P * K / 1e9,
            "acceptance_fraction": acc_frac,
            "finite_lnprob_fraction": finite_frac,
        }
        if not args.no_cpu_baseline and args.dtype == "f64" and world == 1:      # rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
