"""Command line entry point with the reference's contract (vamp_1.0/do_vamp.py:14-60):

    python -m vamp_amd.do_vamp data_file line [--output_folder DIR] [--voigt] [--parallel N] [--conv_attempts N]

``data_file`` is one spectrum, or (with --parallel) a folder of ``spectrum_*.h5`` files that are
distributed over the visible GPUs, one worker process per GPU (the reference's mp.Pool branch,
do_vamp.py:64-96, calls an undefined function and never ran).  New flags: --walkers, --iterations,
--burn, --thin, --seed, --gpus.
"""
import argparse
import glob
import os
import sys


def fit_one(path, args, device=0):
    from .vpspectrum import VPspectrum
    spec = VPspectrum(args.line, path, args.output_folder, voigt=args.voigt, chi_limit=1.5, mcmc_cov=False,
                      get_mcmc_err=True, convergence_attempts=args.conv_attempts, nwalkers=args.walkers,
                      iterations=args.iterations, thin=args.thin, burn=args.burn, seed=args.seed)
    return spec.fit_spectrum(batched=args.batched)


def _worker(rank, files, args):
    os.environ["HIP_VISIBLE_DEVICES"] = str(rank)         # before any HIP call in this process
    for f in files:
        fit_one(f, args, device=0)


def main(argv=None):
    p = argparse.ArgumentParser(description="Voigt Automatic MCMC Profiles on MI355X")
    p.add_argument("data_file", help="spectrum file (HDF5 / npz / 4-column text), or a folder with --parallel")
    p.add_argument("line", type=float, help="rest wavelength of the absorption line [Angstrom]")
    p.add_argument("--output_folder", default=None, help="folder for plots and result files")
    p.add_argument("--voigt", action="store_true", help="fit Voigt profiles (default: Gaussian)")
    p.add_argument("--parallel", type=int, default=1, help="number of worker processes (one GPU each)")
    p.add_argument("--conv_attempts", type=int, default=10, help="fit attempts per region")
    p.add_argument("--walkers", type=int, default=None)
    p.add_argument("--iterations", type=int, default=3000)
    p.add_argument("--burn", type=int, default=300)
    p.add_argument("--thin", type=int, default=15)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--batched", action="store_true",
                   help="fit all regions of a spectrum together (one kernel launch per half-step for the whole spectrum)")
    args = p.parse_args(argv)
    if args.output_folder is not None:
        os.makedirs(args.output_folder, exist_ok=True)
        if not args.output_folder.endswith(os.sep):
            args.output_folder += os.sep
    if args.parallel <= 1 or os.path.isfile(args.data_file):
        fit_one(args.data_file, args)
        return 0
    files = sorted(glob.glob(os.path.join(args.data_file, "spectrum_*.h5")) +
                   glob.glob(os.path.join(args.data_file, "spectrum_*.npz")))
    if not files:
        sys.exit("no spectrum_* files in " + args.data_file)
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, files[r::args.parallel], args)) for r in range(args.parallel)]
    for pr in procs:
        pr.start()
    rc = 0
    for pr in procs:
        pr.join()
        rc = rc or pr.exitcode
    return rc


if __name__ == "__main__":
    sys.exit(main())
