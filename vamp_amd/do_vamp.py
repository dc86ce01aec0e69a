"""Command line entry point with the reference's contract (vamp_1.0/do_vamp.py:14-60):

    python -m vamp_amd.do_vamp data_file line [--output_folder DIR] [--voigt] [--parallel N] [--conv_attempts N]

``data_file`` is one spectrum, or a folder of ``spectrum_*.h5`` / ``spectrum_*.npz`` files.  With
``--parallel N`` the files of a folder are dealt to N worker processes (the reference's mp.Pool
branch, do_vamp.py:64-96, calls an undefined function and never ran); worker r runs on GPU
``r % gpus`` and pins it with HIP_VISIBLE_DEVICES before anything in that process touches HIP.
Spectra are independent: no communication.  New flags: --walkers, --iterations, --burn, --thin,
--seed, --batched, --gpus (GPUs to use; default: the GPUs this process can see), --dtype {f64,f32}
(per-pixel arithmetic; $VAMP_DTYPE overrides the default), --backend hip ($VAMP_BACKEND).
"""
import argparse
import glob
import importlib
import os
import sys


def visible_gpus():
    """Number of GPUs a worker may be pinned to, found WITHOUT initialising HIP in this process (the
    workers are spawned afterwards): HIP_VISIBLE_DEVICES if set, else the KFD topology nodes that
    have SIMDs (CPUs are nodes too, with simd_count 0); at least 1."""
    env = os.environ.get("HIP_VISIBLE_DEVICES")
    if env is not None and env.strip() != "":
        return max(1, len([t for t in env.split(",") if t.strip() != ""]))
    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(prop):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
        except (OSError, ValueError):
            pass
    return max(1, n)


def spectrum_files(folder):
    return sorted(glob.glob(os.path.join(folder, "spectrum_*.h5")) + glob.glob(os.path.join(folder, "spectrum_*.npz")))


def plan_workers(files, parallel, gpus):
    """[(device, [files...]), ...]: at most ``parallel`` workers, none without a file; worker r
    takes files r, r + n, r + 2n, ... and GPU r % gpus."""
    n = max(1, min(int(parallel), len(files)))
    return [(r % max(1, int(gpus)), files[r::n]) for r in range(n)]


def perf_record(spec, path, seconds, batched):
    """One JSON-able record per fitted spectrum (SURVEY section 5 "Metrics / logging": the reference only
    prints; its one timing is VPfit.fit_time, vpfits.py:392-395, kept here per region)."""
    import numpy as np
    regs = getattr(spec, "regions", [])
    chi = np.array([r.best_chi_squared for r in regs], dtype=float)
    fit_s = [float(getattr(r.fit, "fit_time", 0.0) or 0.0) for r in regs]
    return {"spectrum": os.path.basename(str(path)), "regions": len(regs), "lines": int(sum(r.n for r in regs)),
            "pixels_in_regions": int(sum(r.num_pixels for r in regs)), "seconds": float(seconds), "batched": bool(batched),
            "sampler_seconds_last_fits": float(sum(fit_s)), "median_reduced_chi2": float(np.median(chi)) if chi.size else None,
            "frac_regions_below_chi_limit": float(np.mean(chi < spec.chi_limit)) if chi.size else None,
            "difficult_fit": bool(spec.flux_model.get("difficult_fit", False)), "voigt": bool(spec.voigt),
            "dtype": "f32" if getattr(spec, "dtype", 0) == 1 else "f64"}


def fit_one(path, args, device=0):
    import json
    import time
    from .vpspectrum import VPspectrum
    spec = VPspectrum(args.line, path, args.output_folder, voigt=args.voigt, chi_limit=1.5, mcmc_cov=False,
                      get_mcmc_err=True, convergence_attempts=args.conv_attempts, nwalkers=args.walkers,
                      iterations=args.iterations, thin=args.thin, burn=args.burn, seed=args.seed, dtype=args.dtype, device=device)
    t0 = time.perf_counter()
    params = spec.fit_spectrum(batched=args.batched)
    rec = perf_record(spec, path, time.perf_counter() - t0, args.batched)
    print("vamp_perf " + json.dumps(rec), flush=True)
    if args.output_folder is not None and getattr(spec, "output_filename", None):
        with open(spec.output_filename + "perf.json", "w") as fh:
            json.dump(rec, fh)
    return params


def _worker(device, files, args, fit_name):
    # pin the GPU first: nothing imported so far in this (spawned) process has touched HIP
    visible = [t.strip() for t in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if t.strip() != ""]
    os.environ["HIP_VISIBLE_DEVICES"] = visible[device] if device < len(visible) else str(device)
    mod, _, fn = fit_name.partition(":")
    fit = getattr(importlib.import_module(mod), fn)
    for f in files:
        fit(f, args, device=0)


def main(argv=None, _fit_name="vamp_amd.do_vamp:fit_one"):
    p = argparse.ArgumentParser(description="Voigt Automatic MCMC Profiles on MI355X")
    p.add_argument("data_file", help="spectrum file (HDF5 / npz / 4-column text), or a folder with --parallel")
    p.add_argument("line", type=float, help="rest wavelength of the absorption line [Angstrom]")
    p.add_argument("--output_folder", default=None, help="folder for plots and result files")
    p.add_argument("--voigt", action="store_true", help="fit Voigt profiles (default: Gaussian)")
    p.add_argument("--parallel", type=int, default=1, help="worker processes for a folder of spectra (worker r on GPU r %% gpus)")
    p.add_argument("--gpus", type=int, default=0, help="GPUs to spread the workers over (default: all visible)")
    p.add_argument("--conv_attempts", type=int, default=10, help="fit attempts per region")
    p.add_argument("--walkers", type=int, default=None)
    p.add_argument("--iterations", type=int, default=3000)
    p.add_argument("--burn", type=int, default=300)
    p.add_argument("--thin", type=int, default=15)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--dtype", choices=["f64", "f32"], default=None,
                   help="per-pixel arithmetic of the device kernels: f64 (the reference's; default) or f32 = Humlicek W4 "
                        "(BASELINE config 5: chi^2 within 1e-3 of fp64); default: $VAMP_DTYPE, else f64")
    p.add_argument("--backend", choices=["hip"], default=os.environ.get("VAMP_BACKEND", "hip"),
                   help="compute backend: hip (libvamp_hip.so on a GPU).  There is no CPU backend in the product: "
                        "oracle/libvamp_cpu.so is test infrastructure")
    p.add_argument("--batched", action="store_true",
                   help="fit all regions of a spectrum together (one kernel launch per half-step for the whole spectrum)")
    args = p.parse_args(argv)
    if args.backend != "hip":            # (argparse does not check a default taken from the environment)
        sys.exit("do_vamp: backend %r is not available: the product runs on libvamp_hip.so only (no CPU fallback)" % args.backend)
    if args.output_folder is not None:
        os.makedirs(args.output_folder, exist_ok=True)
        if not args.output_folder.endswith(os.sep):
            args.output_folder += os.sep
    if os.path.isfile(args.data_file):
        fit_one(args.data_file, args)
        return 0
    files = spectrum_files(args.data_file)
    if not files:
        sys.exit("no spectrum_* files in " + args.data_file)
    if args.parallel <= 1:
        for f in files:                      # one process, one GPU, one spectrum after the other
            fit_one(f, args)
        return 0
    gpus = args.gpus if args.gpus else visible_gpus()
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(dev, fs, args, _fit_name)) for dev, fs in plan_workers(files, args.parallel, gpus)]
    for pr in procs:
        pr.start()
    rc = 0
    for pr in procs:
        pr.join()
        rc = rc or pr.exitcode
    return rc


if __name__ == "__main__":
    sys.exit(main())
