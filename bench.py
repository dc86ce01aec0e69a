#!/usr/bin/env python3
"""Headline benchmark: walker-steps/s of the stretch-move + Voigt log-posterior hot path on the
synthetic 16 384-pixel x 16-component x 65 536-walker region of BASELINE.json (config 4).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
every rank finds RANK / LOCAL_RANK / WORLD_SIZE in its environment.  Started plainly (`python bench.py
--gpus N`, no WORLD_SIZE) this process touches no GPU: it starts that same launcher as a CHILD process,
relays rank 0's JSON line and exits with the child's code.

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


def make_workload(P=16384, K=16, W=65536, seed=20240517, nbz=True, ensemble="truth", width_scale=1.0):
    """SURVEY section 8d synthetic region.  Region-centred pixel coordinates x_i = i - (P-1)/2;
    truth: centroid ~ U(-0.45P, 0.45P), amplitude ~ U(0.2, 3), G_fwhm ~ U(20, 200) px,
    L_fwhm ~ U(1, 20) px (fixed to 10 px in the 3-parameter NBZ form); sigma = 0.01;
    walkers = truth * (1 + 1e-3 N(0,1)), clipped into the priors of vpfits.py:249-297.

    Robustness knobs (the defaults are the headline): ``width_scale`` multiplies the true widths;
    ``ensemble`` = "truth" (above), "dispersed" (a burn-in ensemble: every walker has its own
    centroids ~ U over the region, amplitudes ~ the xexp prior, widths log-uniform over
    [1/4, 4] x the truth range) or "prior" (every parameter drawn from its prior of
    vpfits.py:239-252, 283-297: widths ~ U(0, fwhm_max), i.e. mostly lines wider than the region)."""
    rng = np.random.default_rng(seed)
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    scale = P / 16384.0
    c = rng.uniform(-0.45 * P, 0.45 * P, K)
    A = rng.uniform(0.2, 3.0, K)
    G = rng.uniform(20.0, 200.0, K) * max(scale, 1.0 / 16) * width_scale
    L = rng.uniform(1.0, 20.0, K) * max(scale, 1.0 / 16) * width_scale
    if nbz:
        L = np.full(K, L_FIXED_PIX * max(scale, 1.0 / 16) * width_scale)
    tau = np.zeros(P)
    for k in range(K):
        tau += _voigt_tau(x, A[k], c[k], L[k], G[k])
    noise = np.full(P, 0.01)
    flux = np.exp(-tau) + rng.normal(0.0, 0.01, P)
    fwhm_max = (x[-1] - x[0]) / 2.0 * FWHM_PER_SIGMA
    out = dict(x=x, flux=flux, noise=noise, K=K, P=P, W=W)
    pert = 1.0 + 1e-3 * rng.standard_normal((W, K, 4))
    nat = np.stack([A, c, L, G], axis=1)[None, :, :] * pert          # (W, K, 4) = A, c, L, G
    if ensemble != "truth":
        r2 = np.random.default_rng(seed + 1)
        ws = max(scale, 1.0 / 16) * width_scale
        nat[:, :, 0] = r2.gamma(2.0, 1.0, (W, K))                     # xexp prior: v exp(-v)
        nat[:, :, 1] = r2.uniform(x[0], x[-1], (W, K))
        if ensemble == "prior":
            nat[:, :, 3] = r2.uniform(0.0, fwhm_max, (W, K))
            if not nbz:
                nat[:, :, 2] = r2.uniform(0.0, fwhm_max, (W, K))
        elif ensemble == "dispersed":
            nat[:, :, 3] = np.exp(r2.uniform(np.log(5.0 * ws), np.log(800.0 * ws), (W, K)))
            if not nbz:
                nat[:, :, 2] = np.exp(r2.uniform(np.log(0.25 * ws), np.log(80.0 * ws), (W, K)))
        else:
            raise ValueError("ensemble must be 'truth', 'dispersed' or 'prior'")
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


def _pool_lnprob(args):
    """worker of the named CPU path: numpy + scipy.special.wofz log-posterior of a chunk of walkers"""
    thetas, = args
    return _POOL_VO.log_prob_batch_fast(_POOL_REGION, thetas)


_POOL_VO = None
_POOL_REGION = None


def _pool_init(region_kw):
    global _POOL_VO, _POOL_REGION
    from oracle import vamp_oracle as vo
    _POOL_VO = vo
    _POOL_REGION = vo.Region(**region_kw)


def cpu_reference_path(wl, budget_s=12.0):
    """The CPU path the north star names (SURVEY 8d baseline 1, BASELINE.md Baseline A): the
    log-posterior in numpy + scipy.special.wofz (oracle/vamp_oracle.py: log_prob_batch_fast) and the
    numpy stretch move (stretch_half_step, same counter-based draws), parallelised over walker
    chunks with a process pool on the host cores.  One full step of a bounded sub-ensemble of the
    same workload is timed (pool start-up and the initial log-posteriors are not)."""
    import multiprocessing as mp
    from oracle import vamp_oracle as vo
    # all host cores = every core of this process's affinity mask (VAMP_CPU_WORKERS caps it)
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("VAMP_CPU_WORKERS", "100000")))
    kw = dict(x=wl["x"], flux=wl["flux"], noise=wl["noise"], n_comp=wl["K"], mode=wl["mode"])
    if wl["nbz"] is not None:
        l_fixed, line, x_origin, x_scale = [float(v) for v in wl["nbz"][0]]
        kw.update(l_fixed=l_fixed, line=line, x_origin=x_origin, x_scale=x_scale)
    chunk = 2                                    # walkers per task: 2 x P x K complex temporaries stay in cache
    with mp.get_context("fork").Pool(cores, initializer=_pool_init, initargs=(kw,)) as pool:
        def lnprob(q):
            parts = pool.map(_pool_lnprob, [(q[i:i + chunk],) for i in range(0, q.shape[0], chunk)])
            return np.concatenate(parts)
        lnprob(wl["theta0"][:chunk * cores])                               # warms the workers
        rng = np.random.default_rng(7)

        def one_step(Wc):
            X = np.ascontiguousarray(wl["theta0"][:Wc].copy())
            lnp = lnprob(X)                                                  # initial log-posteriors: not timed
            nacc = 0
            t0 = time.perf_counter()
            perm = rng.permutation(Wc)                                       # emcee: shuffled red/blue membership
            for act, comp in ((perm[:Wc // 2], perm[Wc // 2:]), (perm[Wc // 2:], perm[:Wc // 2])):
                zz = ((2.0 - 1.0) * rng.random(act.size) + 1.0) ** 2 / 2.0  # a = 2
                partner = comp[rng.integers(0, comp.size, act.size)]
                acc, _ = vo.stretch_half_step(X, lnp, act, partner, zz, np.log(rng.random(act.size)), lnprob)
                nacc += int(acc.sum())
            return time.perf_counter() - t0, nacc

        Wc = min(wl["W"], 8 * chunk * cores)
        t, nacc = one_step(Wc)
        while t < budget_s / 4 and Wc < wl["W"]:                             # grow the sample until it is a few seconds of work
            Wc = int(min(wl["W"], Wc * min(8.0, max(2.0, 0.7 * budget_s / t))))
            Wc -= Wc % (2 * chunk)
            t, nacc = one_step(Wc)
    return {"value": Wc / t, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "path": "numpy + scipy.special.wofz log-posterior + numpy stretch move (oracle/vamp_oracle.py), "
                    "multiprocessing pool over walker chunks: the CPU path named by BASELINE.json",
            "sample": f"first {Wc} walkers of the same workload x 1 full step on {cores} processes (one per core of the "
                      f"affinity mask, {len(os.sched_getaffinity(0))} cores), {t:.1f} s; os.cpu_count() = {os.cpu_count()}",
            "accepted": nacc,
            "note": "BASELINE.md section 3 asks for all host cores, so this is one process per core of the affinity mask (SMT "
                    "threads included).  On the 128-core / 256-thread hosts of these boxes one process per PHYSICAL core is "
                    "faster (1 751-1 760 walker-steps/s with VAMP_CPU_WORKERS=128, round 2) than one per thread (1 485-1 523): "
                    "wofz is bound by the FP units the two threads of a core share"}


def cpu_c_port(wl, budget_s=10.0):
    """Second, labelled figure: the plain-C restatement (oracle/vamp_oracle.c, OpenMP over walkers,
    its own straightforward Re w).  Test infrastructure used as a reported baseline only."""
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
            "path": "oracle/vamp_oracle.c (plain C, OpenMP over walkers)",
            "sample": f"first {Wc} walkers of the same workload x 1 step on {cores} threads, {t:.1f} s"}


def cpu_host_abi(wl, budget_s=8.0):
    """Third figure (BASELINE.md Baseline B): oracle/libvamp_cpu.so, the C ABI of include/vamp_hip.h
    implemented on the host (OpenMP over walkers, the host build of the product's Voigt evaluators,
    the same sampler), driven through the same ctypes wrapper as the GPU library."""
    so = os.path.join(ROOT, "oracle", "libvamp_cpu.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    from vamp_amd import _lib, hip_backend
    ctx = hip_backend.HipContext(lib=_lib.bind(so))
    ctx.set_regions(wl["x"], wl["flux"], wl["noise"], wl["K"], mode=wl["mode"], nbz=wl["nbz"])
    cores = len(os.sched_getaffinity(0))

    def run(Wc):
        blk = hip_backend.default_split_block(Wc)
        ctx.sampler_init(np.ascontiguousarray(wl["theta0"][:Wc]), seed=7, split_block=blk)
        return ctx.run(1, store_chain=False)["seconds"]

    Wc = min(wl["W"], 4 * cores)
    t = run(Wc)
    while t < budget_s / 4 and Wc < wl["W"]:
        Wc = int(min(wl["W"], Wc * min(8.0, max(2.0, 0.7 * budget_s / t))))
        Wc -= Wc % 2
        t = run(Wc)
    ctx.close()
    return {"value": Wc / t, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "path": "oracle/libvamp_cpu.so: the C ABI of include/vamp_hip.h on the host (OpenMP, host build of voigt_math.hpp)",
            "sample": f"first {Wc} walkers of the same workload x 1 step on {cores} threads, {t:.1f} s"}


def cpu_baseline(wl):
    """`value` is the path BASELINE.json names; the two compiled ports are reported beside it."""
    out = cpu_reference_path(wl)
    out["c_port"] = cpu_c_port(wl)
    out["host_abi"] = cpu_host_abi(wl)
    return out


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


# VALU issue roof of one MI355X: 256 CUs x 4 SIMDs x 2.4 GHz x 16 lanes per clock in fp64 (a wave instruction holds
# its SIMD for 4 cycles), 32 lanes per clock in fp32 (2 cycles; one wavefront alone can issue only every 4, so the
# fp32 roof needs >= 2 ready wavefronts per SIMD) -- MI355X_MICROARCH.md: 78.6 / 157.3 TFLOP/s vector peaks
VALU_PEAK_LANE_INSTR = 256 * 4 * 16 * 2.4e9
VALU_PEAK_LANE_INSTR_F32 = 256 * 4 * 32 * 2.4e9
FP64_PEAK_TFLOPS = 2.0 * VALU_PEAK_LANE_INSTR / 1e12          # 78.6 (an FMA counts two)


def committed_pmc(P, K, W, D, world, dtype, workload="headline"):
    """Counters of THIS command from the committed PMC passes (profiles/pmc_traffic*.json; tools/pmc.sh runs them
    as separate rocprofv3 --pmc passes, tools/make_pmc_json.py condenses them), or None when no file matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic*.json"))):
        try:
            pj = json.load(open(path))
            c = pj["config"]
            if c.get("workload", "headline") != workload or c["dtype"] != dtype or c["n_gpus"] != world or c["walkers"] != W:
                continue
            if workload == "headline" and (c["pixels"], c["components"], c["ndim"]) != (P, K, D):
                continue
            return pj
        except (OSError, KeyError, ValueError):
            continue
    return None


# builder-counted arithmetic of ONE direct evaluation (DESIGN.md section 3 "flop count"):
# F_w  = flops of sqrt(pi) Re w(z) incl. forming z and the tau FMA, averaged over this workload's mix of
#        branches when every (pixel, line) pair is evaluated directly (build e: 45.6 VALU instructions per
#        pair, 36 of them FMAs) ; F_px = per-pixel epilogue (exp(-tau) 17 FMA + 5, residual, chi^2 FMA)
F_W_FLOPS, F_PX_FLOPS = 82.0, 46.0


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process
    (`python -m torch.distributed.run`, one rank per GPU) and relay rank 0's line.  This parent never
    initialises the GPU (no torch.cuda, no vamp_amd call) and never execs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no WORLD_SIZE in the environment: starting %d ranks: %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    n_lines = 0
    for ln in child.stdout:
        if ln.lstrip().startswith("{"):           # rank 0's JSON line goes to stdout, anything else to stderr
            sys.stdout.write(ln)
            sys.stdout.flush()
            n_lines += 1
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    if rc == 0 and n_lines != 1:
        print("bench.py: the ranks printed %d JSON lines (expected one)" % n_lines, file=sys.stderr)
        rc = 1
    return rc


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
    ap.add_argument("--ensemble", choices=["truth", "dispersed", "prior"], default="truth",
                    help="initial walkers: the truth ball of SURVEY 8d (default, the headline); 'dispersed' = a burn-in "
                         "ensemble (own centroids, amplitudes and log-uniform widths per walker); 'prior' = drawn from the priors")
    ap.add_argument("--width-scale", type=float, default=1.0, help="multiply the true line widths (G and L) of the workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="do not record the chain inside the timed region")
    ap.add_argument("--sustain-seconds", type=float, default=10.0,
                    help="after the K timed steps, a second run of about this many seconds (0 = skip); long enough for a "
                         "driver that samples GPU activity every few seconds to land inside the kernel")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL communicator and run the exchange even with one rank (rehearsal)")
    ap.add_argument("--exchange-parts", default="auto",
                    help="pieces per half-step of the overlapped exchange: a number, or 'auto' = time 1 and 2 during warm-up "
                         "and keep the faster (every rank takes the same decision from the max over ranks)")
    args = ap.parse_args()

    # the host driver of these boxes only supports dmabuf IPC: without this RCCL's intra-node handles fail
    # (hipIpcGetMemHandle: invalid argument).  Already exported on the boxes; kept for any other launcher.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "VAMP_BENCH_DEVICE" in os.environ:          # rehearsal knob: several ranks on one GPU (host logic only)
        local_rank = int(os.environ["VAMP_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    wl = make_workload(P=args.pixels, K=args.components, W=args.walkers, nbz=(args.param == "nbz3"),
                       ensemble=args.ensemble, width_scale=args.width_scale)
    P, K, W, D = wl["P"], wl["K"], wl["W"], wl["D"]

    # host baseline first: its worker pool forks before this process has touched the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f64":
        cpu = cpu_baseline(wl)

    import torch
    import vamp_amd

    quiet = _StdoutToStderr()
    quiet.__enter__()
    dist = None
    if world > 1 or args.force_dist:
        # torch.distributed is the bootstrap channel only (communicator id, barriers, the final max):
        # gloo on the host.  The per-half-step exchange is RCCL inside libvamp_hip.so.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dtype = vamp_amd.F64 if args.dtype == "f64" else vamp_amd.F32

    def new_ctx():
        c = vamp_amd.HipContext(device=local_rank, dtype=dtype)
        c.set_regions(wl["x"], wl["flux"], wl["noise"], K, mode=wl["mode"], nbz=wl["nbz"])
        return c

    ctx = new_ctx()

    from vamp_amd.ensemble import ShardedEnsemble
    exchange_label, exchange_kind = "none", "none"
    want = "rccl" if dist is not None else "none"

    class CommunicatorFailed(Exception):
        """VampError(-3) out of the ShardedEnsemble CONSTRUCTOR: the one failure every rank is known to see together
        (vamp_amd/ensemble.py: _join_communicator).  Only this leads to the fall-back below; a -3 from a later
        collective (run_dev in the parts probe) may hit one rank alone and must end the run, not desynchronise it."""

    def build(parts, kind=want):
        try:
            return ShardedEnsemble(ctx, wl["theta0"], seed=20240517, dist=dist, exchange=kind, parts=parts,
                                   exchange_single_rank=args.force_dist)
        except vamp_amd._lib.VampError as e:
            if e.code != -3 or dist is None or kind != "rccl":
                raise
            err = CommunicatorFailed(str(e))
            err.stuck = getattr(e, "stuck", False)
            raise err from e

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    # pieces per half-step: a number, or the faster of 1 and 2 measured here (max over ranks, so every rank agrees)
    parts_tried = None
    comm_error = None
    ens = None
    try:
        if dist is None:
            ens = build(1)
        elif args.exchange_parts != "auto":
            ens = build(int(args.exchange_parts))
        else:
            from vamp_amd.hip_backend import default_split_block
            chunks = W // default_split_block(W, world)
            cands = [p for p in (1, 2) if chunks % (world * p) == 0]
            parts_tried = {}
            for p in cands:
                ens = build(p)
                n_try = max(3, args.warmup)
                ens.run_dev(n_try)
                sync_all()
                t0 = time.perf_counter()
                ens.run_dev(n_try)
                sync_all()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                parts_tried[p] = float(t[0]) / n_try * 1e3
            best = min(parts_tried, key=parts_tried.get)
            ens = build(best)
    except CommunicatorFailed as e:
        # ShardedEnsemble raises this on EVERY rank together (vamp_amd/ensemble.py: _join_communicator), so all
        # ranks arrive here: the host-staged exchange -- same kernels, rows through pinned host memory and gloo --
        # is slow and loudly labelled, but the scaling run still produces a correct line instead of none
        ens, comm_error = None, str(e)
        if rank == 0:
            print("bench.py: RCCL communicator failed (%s); FALLING BACK to the host-staged gloo exchange" % comm_error,
                  file=sys.stderr, flush=True)
        if not getattr(e, "stuck", False):
            ctx.close()
        ctx = new_ctx()
        ens = build(1, kind="gloo_host")
        exchange_label = "FALLBACK: host-staged gloo all-gather (RCCL communicator failed: %s)" % comm_error
        exchange_kind = "gloo_host FALLBACK"
    rccl_ranks = rccl_queried = rccl_library = None
    if dist is not None and comm_error is None:
        _, rccl_ranks, rccl_queried = ctx.comm_info()
        exchange_kind = "rccl"
        exchange_label = f"in-library RCCL all-gather of the active colour, {ens.parts} piece(s) per half-step"
        try:
            rccl_library = vamp_amd.hip_backend.comm_library()      # the shared object whose ncclAllGather carried the bytes
        except vamp_amd._lib.VampError as e:
            rccl_library = "unknown (%s)" % e
    # which physical device every rank ran on: a scaling line over N ranks must name N different devices
    props = torch.cuda.get_device_properties(local_rank)
    my_dev = {"rank": rank, "device_index": local_rank, "name": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
    device_of_rank = [my_dev]
    if dist is not None and world > 1:
        box = [None] * world
        dist.all_gather_object(box, my_dev)
        device_of_rank = box
    # rehearsal knobs: with either of them a line over N ranks does NOT measure N GPUs over xGMI, and says so
    knobs = {k: os.environ[k] for k in ("VAMP_RCCL_LIB", "VAMP_BENCH_DEVICE") if k in os.environ}
    distinct = len({(d["uuid"] or d["pci_bus_id"] or d["device_index"]) for d in device_of_rank})
    rehearsal = bool(knobs) or (world > 1 and distinct < world)
    own = ens.own_count
    host_staged = ens.exchange == "gloo_host"

    def run_steps(n, thin=1, chain_ptr=None):
        if host_staged:
            ens.step(n)                   # stepped from Python, no device-resident chain in this mode
        else:
            ens.run_dev(n, thin=thin, chain_ptr=chain_ptr)

    # chain storage (device resident, torch = allocator): every kept step is one device-to-device copy
    # of the state on the stream the kernels run on, inside the timed region
    chain = None if (args.no_chain or host_staged) else torch.empty((args.steps, W * D), dtype=torch.float64, device=dev)

    run_steps(args.warmup)
    sync_all()
    quiet.__exit__()
    ctx.kernel_timing(True)
    t0 = time.perf_counter()
    run_steps(args.steps, chain_ptr=None if chain is None else chain.data_ptr())
    sync_all()
    dt = time.perf_counter() - t0
    x_ms, x_n = ctx.exchange_timing()
    k_ms, k_n = ctx.kernel_timing(False)
    if dist is not None:
        # the max over ranks NOW: the length of the second run is derived from it, and every rank must step the
        # same number of times (a rank that ran one step more would wait in the exchange for ever)
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # a longer second run (clocks and caches settled; 20 steps are 0.15 s): same loop, the chain thinned
    # into the same buffer
    sustained = None
    if args.sustain_seconds > 0:
        n_s = max(args.steps, int(np.ceil(args.sustain_seconds / (dt / args.steps))))
        thin = int(np.ceil(n_s / args.steps))
        n_s -= n_s % thin
        sync_all()
        ctx.kernel_timing(True)
        t0 = time.perf_counter()
        run_steps(n_s, thin=thin, chain_ptr=None if chain is None else chain.data_ptr())
        sync_all()
        dts = time.perf_counter() - t0
        ctx.exchange_timing()
        s_ms, s_n = ctx.kernel_timing(False)
        sustained = {"steps": n_s, "seconds": dts, "chain_thin": thin, "avg_launch_ms": s_ms / max(1, s_n), "launches": s_n}

    if dist is not None and sustained:
        t = torch.tensor([sustained["seconds"]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sustained["seconds"] = float(t[0])

    # sanity: the ensemble is alive (some proposals accepted, lnprob finite)
    _, lnp, nacc, n_done = ctx.get_state()
    acc_frac = float(nacc[ens.own_mask].mean()) / max(1, n_done)
    finite_frac = float(np.isfinite(lnp).mean())

    if rank == 0:
        value = W * args.steps / dt
        s = 8 if args.dtype == "f64" else 4
        b_alg = algorithmic_bytes_per_walker_step(P, D, s)
        per_launch_units = own // (2 * ens.parts)       # walker-steps of one half-step launch on this rank
        avg_ms = k_ms / max(1, k_n)
        traffic = valu_instr = valu_busy = pmc_src = None
        pj = committed_pmc(P, K, W, D, world, args.dtype) if (dist is None and args.ensemble == "truth" and args.width_scale == 1.0) else None
        if pj is not None:
            traffic = (2.0 * pj["FETCH_SIZE_KB_per_launch"] + pj["WRITE_SIZE_KB_per_launch"]) * 1024.0
            valu_instr = pj.get("SQ_INSTS_VALU_per_launch")
            valu_busy = pj.get("valu_busy_frac_pmc")
            pmc_src = pj.get("source")
        valu_peak = VALU_PEAK_LANE_INSTR if args.dtype == "f64" else VALU_PEAK_LANE_INSTR_F32
        achieved = per_launch_units * b_alg / (avg_ms * 1e-3) / 1e9 if k_n else None
        flop_ws = P * (K * F_W_FLOPS + F_PX_FLOPS)
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
                       "chain_recorded": chain is not None,
                       "ensemble": args.ensemble, "width_scale": args.width_scale,
                       "exchange": exchange_label},
            # contract form: ALGORITHMIC bytes of the launch / its HIP-event duration, against the HBM peak.
            # The kernel is not HBM-bound (the spectrum is shared by all walkers and stays in L2: see
            # `traffic`); its binding roof is fp64 VALU issue, reported in `valu` below.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                         "achieved_is": "algorithmic GB/s (SURVEY 8d bytes per walker-step x walker-steps per launch / launch time)",
                         "traffic_is": "HBM bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB, from the committed PMC passes "
                                       "of this command (not measured in this run: PMC needs separate passes)",
                         "traffic_source": pmc_src,
                         "measured_hbm_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and k_n) else None,
                         "algorithmic_bytes_per_launch": per_launch_units * b_alg,
                         "kernel": "k_half_step", "avg_launch_ms": avg_ms, "launches": k_n,
                         "alg_bytes_per_walker_step": b_alg, "walker_steps_per_launch": per_launch_units,
                         "binding_roof": "%s VALU issue" % ("fp64" if args.dtype == "f64" else "fp32"),
                         "valu": None if not (valu_instr and k_n) else {
                             "peak_lane_instr_per_s": valu_peak,
                             "executed_lane_instr_per_s": valu_instr * 64 / (avg_ms * 1e-3),
                             "valu_issue_frac": valu_instr * 64 / (avg_ms * 1e-3) / valu_peak,
                             "valu_issue_frac_is": "against the %s roof (%d lanes per clock and SIMD)%s" % (
                                 args.dtype, 16 if args.dtype == "f64" else 32,
                                 "" if args.dtype == "f64" else "; the kernel's fp64 and integer instructions (staging, table build, "
                                 "classification, the far-field transform, chi^2 accumulation) issue at half that rate, so 1.0 is not reachable"),
                             "valu_busy_frac_pmc": valu_busy if args.dtype == "f64" else None,
                             "valu_busy_is": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), both from the "
                                             "committed PMC passes: the share of SIMD cycles with a VALU instruction in flight "
                                             "(a little above the issue fraction: reciprocals and integer multiplies take more "
                                             "than one 4-cycle pass)",
                             "formula": "SQ_INSTS_VALU (wave instructions per launch, committed PMC pass) x 64 lanes / launch time "
                                        "/ (256 CUs x 4 SIMDs x %d lanes/clk x 2.4 GHz)" % (16 if args.dtype == "f64" else 32)},
                         "flops": None if args.dtype != "f64" else {"F_w": F_W_FLOPS, "F_px": F_PX_FLOPS, "flop_per_walker_step": flop_ws,
                                   "nominal_TFLOPs": value * flop_ws / 1e12 / world, "peak_TFLOPs": FP64_PEAK_TFLOPS,
                                   "frac_nominal": value * flop_ws / 1e12 / world / FP64_PEAK_TFLOPS,
                                   "note": "NOMINAL: P (K F_w + F_px) per walker-step as if every (pixel, line) pair were "
                                           "evaluated directly; the far-field interpolant and the Taylor tables carry out "
                                           "~30 % of that arithmetic, so the nominal fraction can exceed 1"}},
            # how many ranks the exchange really spanned, and what it cost (rank 0's HIP events; half-step = one colour)
            "rccl_ranks": rccl_ranks,
            # true: ranks shared a device and / or a stand-in carried the exchange (VAMP_BENCH_DEVICE, VAMP_RCCL_LIB): the
            # code path of an N-GPU run, NOT a measurement of one
            "rehearsal": rehearsal,
            "rehearsal_knobs": knobs,
            "device_of_rank": device_of_rank,
            "exchange": None if dist is None else {
                "kind": exchange_kind, "parts": ens.parts, "rccl_ranks": rccl_ranks,
                "rccl_library": rccl_library,
                "rccl_library_is": "dladdr of the ncclAllGather the library bound (vamp_comm_library)",
                "distinct_devices": distinct,
                "rccl_ranks_is": None if rccl_ranks is None else (
                    "ncclCommCount of the library's communicator" if rccl_queried else "the world the communicator was created with"),
                "bytes_per_rank_per_half_step": (own // 2) * (D + 1) * 8,
                "kernel_ms_per_half_step": k_ms / (2 * args.steps),
                "exchange_ms_per_half_step": x_ms / (2 * args.steps),
                "wall_ms_per_half_step": dt / (2 * args.steps) * 1e3,
                # share of the exchange time that ran beside a kernel: 1 - (wall - kernel) / exchange
                "overlap_frac": None if not x_n or x_ms <= 0 else
                    max(0.0, min(1.0, 1.0 - (dt * 1e3 - k_ms) / x_ms)),
                "is": "HIP events of rank 0: kernels on the compute stream, ncclAllGather + scatter on the stream they run on "
                      "(the communication stream when parts > 1); wall = the driver-visible time (max over ranks, includes the "
                      "chain copy of every step)",
                "exchanges_timed": x_n,
                "parts_tried_ms_per_step": parts_tried},
            "nominal_faddeeva_gevals_per_s": value * P * K / 1e9,
            "acceptance_fraction": acc_frac,
            "finite_lnprob_fraction": finite_frac,
        }
        if sustained:
            sustained["value"] = W * sustained["steps"] / sustained["seconds"]
            sustained["ms_per_step"] = sustained["seconds"] / sustained["steps"] * 1e3
            line["sustained"] = sustained
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
