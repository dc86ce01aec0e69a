"""CPU tests (-m "not gpu") of the drop-in boundary through the HOST implementation of the C ABI
(oracle/libvamp_cpu.so: the same header, include/vamp_hip.h, implemented with OpenMP and the host
build of the Voigt evaluators).  Test infrastructure: the library is bound explicitly here
(``_lib.bind``); vamp_amd itself never loads it.

What this buys without a GPU: the ctypes layer, argument checks and error codes, the call-order
rules, the MAP search, the sampler's draws / sharding / pack-and-scatter arithmetic and the VPfit
facade all run for real, against the same oracle and golden vectors as the GPU parity tests -- the
parity tests of tests/test_gpu_parity.py are reused verbatim with this context."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import vamp_oracle as vo

import test_gpu_parity as gp
import test_gpu_vpfit as gv

# VAMP_CPU_SO: another build of the same library (tests/test_sanitizers.py points it at the ASan/UBSan build)
CPU_SO = os.environ.get("VAMP_CPU_SO") or os.path.join(ROOT, "oracle", "libvamp_cpu.so")


@pytest.fixture(scope="module")
def cpu_lib():
    if not os.path.exists(CPU_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    from vamp_amd import _lib
    return _lib.bind(CPU_SO)


@pytest.fixture
def cpu_ctx(cpu_lib):
    import vamp_amd
    ctx = vamp_amd.HipContext(lib=cpu_lib)
    ctx.packing_request = 64
    yield ctx
    ctx.close()


def test_host_library_exports_the_whole_header(cpu_lib):
    import subprocess
    from test_abi import _header_functions
    out = subprocess.check_output(["nm", "-D", "--defined-only", CPU_SO], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T vamp_" in l)
    assert exported == _header_functions()
    assert cpu_lib.vamp_version() == 4


@pytest.mark.parametrize("fn", [gp.test_device_wofz_matches_scipy_and_mpmath, gp.test_lnprob_matches_golden,
                                gp.test_line_records_and_prior_match_oracle, gp.test_lnprob_include_norm_and_bounds,
                                gp.test_multi_region_batch_matches_single, gp.test_model_all_equals_model_region_by_region,
                                gp.test_stretch_injected_draws_parity, gp.test_smallest_shapes,
                                gp.test_sampler_resume_and_thin, gp.test_sampler_multi_region_matches_oracle,
                                gp.test_sampler_sd_mode_and_acceptance, gp.test_map_all_follows_scipy_fmin,
                                gp.test_regions_of_more_than_16_lines],
                         ids=lambda f: f.__name__)
def test_gpu_parity_test_through_the_host_abi(fn, cpu_ctx):
    fn(cpu_ctx)


@pytest.mark.parametrize("block", [8, 16])
def test_philox_trajectory_through_the_host_abi(cpu_ctx, block):
    gp.test_stretch_philox_trajectory_parity(cpu_ctx, block)


def test_error_codes_and_call_order(cpu_lib, cpu_ctx):
    import vamp_amd
    E = vamp_amd._lib.VampError
    g = load_golden("stretch_traj.npz")
    z4, out1 = np.zeros(4), np.zeros(1)
    dp = lambda a: a.ctypes.data_as(vamp_amd._lib.c_double_p)
    assert cpu_lib.vamp_lnprob(cpu_ctx._h, 0, 1, dp(z4), dp(out1), None) == -5          # before set_regions
    assert b"vamp_set_regions first" in cpu_lib.vamp_last_error()
    with pytest.raises(E) as e:
        vamp_amd.HipContext(dtype=vamp_amd.F32, wofz_kind=0, lib=cpu_lib)
    assert e.value.code == -1
    for bad in (np.r_[g["x"][:5], g["x"][3], g["x"][6:]], np.r_[g["x"][:4], np.nan, g["x"][5:]]):
        with pytest.raises(E) as e:
            cpu_ctx.set_regions(bad, g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
        assert e.value.code == -1
    cpu_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    with pytest.raises(E) as e:
        cpu_ctx.half_step(0)                                  # before sampler_init
    assert e.value.code == -5
    with pytest.raises(E) as e:
        cpu_ctx.sampler_init(g["X0"][:15], seed=1, split_block=5)
    assert e.value.code == -1
    cpu_ctx.sampler_init(g["X0"], seed=1, split_block=8)
    with pytest.raises(E):                                    # partner inside the active set
        cpu_ctx.half_step_ext([0, 1], [1, 2], [1.0, 1.0], [0.0, 0.0])
    with pytest.raises(E) as e:                               # W/split_block = 2 chunks cannot feed 4 ranks
        cpu_ctx.sampler_set_shard_parts(0, 4, 1)
    assert e.value.code == -1
    cpu_ctx.sampler_set_shard_parts(1, 2, 1)
    with pytest.raises(E) as e:                               # a shard without a communicator is stepped by the host
        cpu_ctx.run(1)
    assert e.value.code == -5
    th = g["X0"].copy()
    th[0, 0] = np.nan
    th[1, 3] = np.inf
    out = cpu_ctx.lnprob(th)
    assert out[0] == -np.inf and out[1] == -np.inf and np.isfinite(out[2:]).all()


def test_pack_and_scatter_arithmetic(cpu_ctx):
    """The active-colour exchange of a 2-shard context (vamp_sampler_pack_get / scatter_put): same
    assertions as the GPU test of the scatter kernel."""
    g = load_golden("stretch_traj.npz")
    region = vo.Region(x=g["x"], flux=g["flux"], noise=g["noise"], n_comp=1, mode=vo.MODE_VOIGT4)
    rng = np.random.default_rng(18)
    X0 = np.stack([rng.uniform(0.3, 1.5, 64), rng.uniform(-4, 4, 64), rng.uniform(0.5, 3, 64), rng.uniform(2, 8, 64)], 1)
    cpu_ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vo.MODE_VOIGT4)
    cpu_ctx.sampler_init(X0, seed=5, split_block=16)
    (b, e), = cpu_ctx.sampler_set_shard_parts(1, 2, 1)
    assert (b, e) == (32, 64)
    cpu_ctx.half_step_part(0, 0)
    mine = cpu_ctx.pack_get(0)
    X1, lnp1, _, _ = cpu_ctx.get_state()
    red, blue = vo.split_tables(5, 0, 64, 16)
    assert np.array_equal(mine[:, :-1], X1[red[16:]]) and np.array_equal(mine[:, -1], lnp1[red[16:]])
    foreign = np.arange(16 * 5, dtype=np.float64).reshape(16, 5) + 1000.0
    cpu_ctx.scatter_put(0, np.concatenate([foreign, -mine]))
    X2, lnp2, _, _ = cpu_ctx.get_state()
    assert np.array_equal(X2[red[:16]], foreign[:, :-1]) and np.array_equal(lnp2[red[:16]], foreign[:, -1])
    keep = np.ones(64, dtype=bool)
    keep[red[:16]] = False
    assert np.array_equal(X2[keep], X1[keep]) and np.array_equal(lnp2[keep], lnp1[keep])


def test_single_rank_exchange_rehearsal(cpu_ctx):
    """comm of one rank + pieces: the call sequence of the production multi-GPU path, same chain"""
    from vamp_amd.ensemble import ShardedEnsemble
    g = load_golden("stretch_traj.npz")
    cpu_ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
    cpu_ctx.sampler_init(g["X0"], seed=99, split_block=4)
    ref = cpu_ctx.run(5)
    import vamp_amd
    with vamp_amd.HipContext(lib=cpu_ctx._lib) as ctx:
        ctx.set_regions(g["x"], g["flux"], g["noise"], 1, mode=vo.MODE_VOIGT4)
        # comm id comes from the bound library: patch the module-level helper for this context
        ctx.comm_init_rank(b"vamp-cpu" + bytes(120), 0, 1)
        ctx.sampler_init(g["X0"], seed=99, split_block=4)
        ctx.sampler_set_shard_parts(0, 1, 2)
        res = ctx.run(5)
        assert np.array_equal(res["chain"], ref["chain"]) and np.array_equal(res["n_accept"], ref["n_accept"])


@pytest.mark.parametrize("voigt,n", [(False, 1), (True, 2)])
def test_vpfit_facade_on_the_host_abi_matches_oracle(cpu_lib, voigt, n):
    """VPfit.find_bic / chain_covariance through ctypes and the host ABI against the oracle-backed
    context: the GPU test of tests/test_gpu_vpfit.py with the host library in place of the HIP one."""
    import vamp_amd
    from oracle_ctx import OracleContext
    from vamp_amd.vpfits import VPfit
    nu, flux, noise = gv._hi_region(2)
    g, o = VPfit(seed=3), VPfit(seed=3)
    g._ctx = vamp_amd.HipContext(lib=cpu_lib)
    o._ctx = OracleContext()
    for fit in (g, o):
        fit.nwalkers = 32
        fit.find_bic(nu, flux, n, noise, nu.size - 3 * n, voigt=voigt, iterations=20, thin=1, burn=5)
    assert np.allclose(g._chain_dev, o._chain_dev, rtol=1e-9, atol=1e-12)
    assert np.allclose(g.bic_array, o.bic_array, rtol=1e-9, atol=0) and np.allclose(g.red_chi_array, o.red_chi_array, rtol=1e-9, atol=0)
    assert np.allclose(g.chain_covariance(n, voigt=voigt), o.chain_covariance(n, voigt=voigt), rtol=1e-7, atol=0)
    assert np.allclose(g.total.value, o.total.value, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("voigt", [False, True])
def test_batched_find_bic_on_the_host_abi_matches_oracle(cpu_lib, voigt):
    """vamp_amd.batched.find_bic_batched -- the three repeats of every region as 3 R regions of ONE context
    (vpfits.py:417-428), array-based host side, fit objects built on demand -- through ctypes and the host ABI
    against the same code on the oracle-backed context: the GPU test of tests/test_gpu_vpfit.py with the host
    library in place of the HIP one.  Also the ladder on top of it (vpregion.py:42-91) end to end."""
    import vamp_amd
    from oracle_ctx import OracleContext
    from vamp_amd.batched import BatchedRegionLadder, find_bic_batched
    from vamp_amd.vpregion import VPregion
    regions = [gv._hi_region(i) for i in range(3)]
    ns = [1, 2, 1]
    out = {}
    for name, ctx in (("abi", vamp_amd.HipContext(lib=cpu_lib)), ("oracle", OracleContext())):
        out[name] = find_bic_batched(ctx, regions, ns, voigt=voigt, nwalkers=32, iterations=20, thin=1, burn=5, seed=17)
        assert ctx.n_regions == 9
    for r, (g, o) in enumerate(zip(out["abi"], out["oracle"])):
        assert len(set(g.bic_array)) == 3                                   # every repeat has a chain of its own
        assert np.allclose(g.bic_array, o.bic_array, rtol=1e-9, atol=0), (r, g.bic_array, o.bic_array)
        assert np.allclose(g.red_chi_array, o.red_chi_array, rtol=1e-9, atol=0)
        fg, fo = g.detach().fit(), o.detach().fit()
        assert np.allclose(fg._chain_dev, fo._chain_dev, rtol=1e-9, atol=1e-12) and fg.mcmc.DIC is None
        assert np.isclose(fg.map.BIC, g.bic_array[-1]) and np.allclose(fg.total.value, fo.total.value, rtol=1e-9, atol=1e-300)
        assert set(fg.mcmc.stats()) >= {"xexp_0", "est_centroid_0", "est_sigma_0", "sd"}
    if not voigt:
        regs = [VPregion(nu, f, n, voigt=False, nwalkers=32, seed=5) for nu, f, n in regions[1:]]
        BatchedRegionLadder(regs, nwalkers=32, iterations=30, thin=1, burn=10, seed=3, verbose=False, ctx=vamp_amd.HipContext(lib=cpu_lib)).run()
        for reg in regs:
            assert 1 <= reg.n <= 6 and len(reg.fit.estimated_profiles) == reg.n and len(reg.fit.bic_array) == 3
            assert reg.fit.total.value.shape == reg.flux_array.shape and np.isfinite(reg.fit.map.BIC)


def test_host_plan_launch_classes_shards_and_grids(cpu_lib, cpu_ctx):
    """csrc/host_plan.hpp -- the launch-plan arithmetic of libvamp_hip.so (launch classes, shard and piece bounds,
    exchange buffer sizes, wavefronts per region of packed launches, the region -> XCD mapping, the resident loop's
    policy) -- through the host build of the ABI, which compiles the same header (and runs it under ASan / UBSan in
    tests/test_sanitizers.py)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    # launch classes of a spectrum-like context: blends, one- and two-line regions, the rest, a region of 20 lines
    shapes = [(40, 1), (200, 4), (30, 5), (512, 8), (513, 8), (95, 3), (96, 3), (60, 2), (150, 20)] + [(20, 1)] * 8     # mean region <= 128 px
    xs = [np.arange(P, dtype=np.float64) for P, _ in shapes]
    cpu_ctx.set_packing(0)
    cpu_ctx.set_regions(xs, [np.ones(P) for P, _ in shapes], [np.ones(P) for P, _ in shapes], [K for _, K in shapes], mode=1)
    kinds, n_cls = cpu_ctx.region_classes()
    assert kinds == [3, 1, 0, 1, 0, 0, 1, 3, 4] + [3] * 8 and n_cls == 4
    short = [i for i in range(len(shapes)) if shapes[i][1] <= 8]
    sub = lambda seq: [seq[i] for i in short]
    cpu_ctx.set_regions(sub(xs), [np.ones(P) for P, _ in sub(shapes)], [np.ones(P) for P, _ in sub(shapes)], [K for _, K in sub(shapes)], mode=0)
    assert cpu_ctx.region_classes() == ([3, 0, 0, 0, 0, 0, 0, 3] + [3] * 8, 2)          # Gaussian components: no tables, no blend class
    # a few regions of 9 .. 16 lines take the one-walker-per-wavefront class on their own
    mix = [(40, 1)] * 6 + [(60, 9), (200, 16), (120, 4)]
    cpu_ctx.set_regions([np.arange(P, dtype=np.float64) for P, _ in mix], [np.ones(P) for P, _ in mix], [np.ones(P) for P, _ in mix],
                        [K for _, K in mix], mode=1)
    assert cpu_ctx.region_classes() == ([3] * 6 + [2, 2, 1], 3)
    long_x = [np.arange(2048, dtype=np.float64), np.arange(4096, dtype=np.float64)]
    cpu_ctx.set_regions(long_x, [np.ones(x.size) for x in long_x], [np.ones(x.size) for x in long_x], [16, 3], mode=1)
    assert cpu_ctx.region_classes() == ([2, 2], 1)
    cpu_ctx.set_packing(16)
    with pytest.raises(Exception, match="at most 8 components"):
        cpu_ctx.set_regions(xs, [np.ones(P) for P, _ in shapes], [np.ones(P) for P, _ in shapes], [K for _, K in shapes], mode=1)
    cpu_ctx.set_regions(sub(xs), [np.ones(P) for P, _ in sub(shapes)], [np.ones(P) for P, _ in sub(shapes)], [K for _, K in sub(shapes)], mode=1)
    assert cpu_ctx.region_classes() == ([0] * 16, 1)
    cpu_ctx.set_packing(0)
    # shards and pieces: every slot of a half-step belongs to exactly one (rank, piece), in whole split chunks
    x = np.arange(40, dtype=np.float64)
    cpu_ctx.set_regions(x, np.ones(40), np.ones(40), 1, mode=1)
    for W, block, world, parts in ((64, 8, 2, 2), (96, 4, 3, 4), (128, 2, 8, 8), (32, 32, 1, 1)):
        X0 = np.column_stack([rng.uniform(0.5, 1, W), rng.uniform(5, 30, W), rng.uniform(1, 3, W), rng.uniform(2, 5, W)])
        cpu_ctx.sampler_init(X0, seed=1, split_block=block)
        owned = np.zeros(W, dtype=int)
        for rank in range(world):
            for b, e in cpu_ctx.sampler_set_shard_parts(rank, world, parts):
                assert b % block == 0 and e % block == 0 and e - b == W // (world * parts)
                owned[b:e] += 1
            if world > 1:
                assert cpu_ctx.pack_get(0).shape == (W // 2 // (world * parts), 5)
        assert np.all(owned == 1)
    with pytest.raises(Exception, match="multiple of world"):
        cpu_ctx.sampler_set_shard_parts(0, 3, 1)
    # packed launches: wavefronts per region, and the region -> XCD mapping is a bijection that keeps a region on one XCD
    cpu_lib.vampdbg_xcd_map.restype = C.c_longlong
    cpu_lib.vampdbg_xcd_map.argtypes = [C.c_longlong] * 3
    out3 = (C.c_longlong * 3)()
    for n_reg, half_w, subs, wpb in ((361, 8192, 8, 2), (54, 8192, 1, 1), (6, 100, 4, 2), (17, 35, 8, 2), (8, 16, 8, 2)):
        cpu_lib.vampdbg_packed_grid(C.c_longlong(n_reg), C.c_longlong(half_w), subs, wpb, out3)
        wpr, grid, bpr = out3[0], out3[1], out3[2]
        assert wpr == -(-half_w // subs) and grid == -(-n_reg * wpr // wpb) and bpr == (wpr // wpb if wpr % wpb == 0 else 0)
        if bpr:
            mapped = [cpu_lib.vampdbg_xcd_map(b, n_reg, bpr) for b in range(grid)]
            assert sorted(mapped) == list(range(grid))
            for b, lb in enumerate(mapped[:(n_reg - n_reg % 8) * bpr]):
                assert (lb // bpr) % 8 == b % 8                      # region i runs on XCD i % 8
    assert cpu_lib.vampdbg_xcd_map(5, 100, 0) == 5
    # the resident loop's automatic policy: packed short-region classes, every mover in one round
    ok = cpu_lib.vampdbg_resident_class_ok
    assert ok(3, 16, 2, 8, 1) == 1 and ok(3, 17, 2, 8, 1) == 0 and ok(1, 16, 8, 1, 1) == 0 and ok(1, 16, 8, 1, 0) == 1 and ok(0, 16, 0, 4, 0) == 0
