"""Multi-rank tests of the walker-sharded driver (vamp_amd/ensemble.py).

CPU (gloo, world_size 2 and 4): the host logic -- shard ownership, the per-half-step all-gather
layout, rank-independence of the trajectory -- with an oracle-backed stand-in for the device
backend (the stand-in lives here, in tests/, and is never reachable from the product).
GPU (-m gpu): two ranks sharing device 0 with the real HIP kernels and the host-staged gloo
exchange, against a single-rank run of the same kernels.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import vamp_oracle as vo


class OracleBackend:
    """CPU stand-in with the sampler surface of HipContext (same slot ownership rules as
    vamp_sampler_set_shard / k_half_step), built on the oracle's counter-based draws."""

    def __init__(self, region):
        self.region = region
        self.fn = lambda q: vo.log_prob_batch_fast(region, q)

    def sampler_init(self, theta0, seed=0, a=2.0, split_block=None):
        self.X = np.array(theta0, dtype=np.float64)
        self.W, self.D = self.X.shape
        self.lnp = self.fn(self.X)
        self.nacc = np.zeros(self.W, dtype=np.int64)
        self.seed, self.a, self.block = seed, a, split_block
        self.step = 0
        self.slot_begin, self.slot_end = 0, self.W // 2

    def sampler_set_shard(self, rank, world):
        chunks = self.W // self.block
        assert chunks % world == 0
        cpr = chunks // world
        self.slot_begin = rank * cpr * (self.block // 2)
        self.slot_end = (rank + 1) * cpr * (self.block // 2)
        return rank * cpr * self.block, (rank + 1) * cpr * self.block

    def sampler_set_shard_parts(self, rank, world, parts):
        chunks = self.W // self.block
        assert chunks % (world * parts) == 0
        cpp = chunks // (world * parts)
        hb = self.block // 2
        self.part_slots = [((p * (chunks // parts) + rank * cpp) * hb, (p * (chunks // parts) + (rank + 1) * cpp) * hb)
                           for p in range(parts)]
        self.slot_begin, self.slot_end = self.part_slots[0]
        self.shard = (rank, world, parts)
        self.packed = [None] * parts
        self.moved = [None] * parts
        return [(2 * b, 2 * e) for b, e in self.part_slots]

    def half_step_part(self, half, part):
        self.slot_begin, self.slot_end = self.part_slots[part]
        step = self.step
        self._move_slots(half)
        # what vamp_sampler_pack_get returns: the movers' rows (position + lnprob) in slot order
        red, blue = vo.split_tables(self.seed, step, self.W, self.block)
        act = red if half == 0 else blue
        mine = act[self.slot_begin:self.slot_end]
        self.packed[part] = np.concatenate([self.X[mine], self.lnp[mine, None]], axis=1)
        self.moved[part] = act
        if half == 1 and part != len(self.part_slots) - 1:
            self.step = step              # the counter advances after the last piece

    def pack_get(self, part=0):
        return self.packed[part].copy()

    def scatter_put(self, part, rows_all):
        """vamp_sampler_scatter_put: rows of piece `part` from all ranks (slot order) -> walker rows"""
        rank, world, parts = self.shard
        n = self.part_slots[part][1] - self.part_slots[part][0]
        first = part * (self.W // 2 // parts)
        ws = self.moved[part][first:first + world * n]
        self.X[ws] = rows_all[:, :-1]
        self.lnp[ws] = rows_all[:, -1]

    def half_step(self, half):
        """all pieces of this rank's share (vamp_sampler_half_step)"""
        parts = getattr(self, "part_slots", None)
        if parts is None:
            return self._move_slots(half)
        for p in range(len(parts)):
            self.half_step_part(half, p)

    def _move_slots(self, half):
        red, blue = vo.split_tables(self.seed, self.step, self.W, self.block)
        act, comp = (red, blue) if half == 0 else (blue, red)
        sl = slice(self.slot_begin, self.slot_end)
        mine = act[sl]
        n = act.size
        zz, logu, partner = np.empty(mine.size), np.empty(mine.size), np.empty(mine.size, dtype=np.int64)
        for i, w in enumerate(mine):
            z, j, lu = vo.draw_move(self.seed, self.step, half, int(w), n, self.a)
            zz[i], logu[i], partner[i] = z, lu, comp[j]
        acc, _ = vo.stretch_half_step(self.X, self.lnp, mine, partner, zz, logu, self.fn)
        self.nacc[mine[acc]] += 1
        if half == 1:
            self.step += 1

    def get_state(self):
        return self.X.copy(), self.lnp.copy(), self.nacc.copy(), self.step

    def set_state(self, theta, lnprob, step):
        self.X[:] = theta
        self.lnp[:] = lnprob
        self.step = step


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    g = load_golden("stretch_traj.npz")
    region = vo.Region(x=g["x"], flux=g["flux"], noise=g["noise"], n_comp=1, mode=vo.MODE_VOIGT4)
    rng = np.random.default_rng(8)
    W = 32
    X0 = np.stack([rng.uniform(0.3, 1.5, W), rng.uniform(-4, 4, W), rng.uniform(0.5, 3, W), rng.uniform(2, 8, W)], 1)
    return region, X0


def _worker(rank, world, port, use_gpu, out_dir, parts=1, block=8):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from vamp_amd.ensemble import ShardedEnsemble
    dist.init_process_group("gloo", rank=rank, world_size=world)
    region, X0 = _case()
    if use_gpu == "cpu_abi":          # the host implementation of the C ABI (oracle/libvamp_cpu.so), bound explicitly
        import vamp_amd
        backend = vamp_amd.HipContext(lib=vamp_amd._lib.bind(os.path.join(ROOT, "oracle", "libvamp_cpu.so")))
        backend.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
    elif use_gpu:
        import vamp_amd
        backend = vamp_amd.HipContext(device=0)
        backend.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
    else:
        backend = OracleBackend(region)
    ens = ShardedEnsemble(backend, X0, seed=4242, split_block=block, dist=dist, exchange="gloo_host", parts=parts)
    assert ens.own_count == X0.shape[0] // world and len(ens.own_ranges) == parts
    ens.step(6)
    X, lnp, nacc = ens.gather_state()
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), X=X, lnp=lnp, nacc=nacc)
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(world, use_gpu, tmp_path, parts=1, block=8):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, use_gpu, str(tmp_path), parts, block), nprocs=world, join=True)
    return np.load(os.path.join(str(tmp_path), "result.npz"))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_trajectory_is_rank_independent_cpu(world, tmp_path):
    region, X0 = _case()
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, lchain, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=8)
    r = _run_ranks(world, False, tmp_path)
    assert np.array_equal(r["X"], chain[-1])            # bit-identical for every world size
    assert np.array_equal(r["lnp"], lchain[-1])
    assert np.array_equal(r["nacc"], nacc)


@pytest.mark.parametrize("world,parts,block", [(2, 2, 8), (4, 2, 4), (2, 4, 4), (8, 2, 2)])
def test_piecewise_exchange_is_rank_independent_cpu(world, parts, block, tmp_path):
    """A rank's share cut into pieces that are stepped and exchanged one after the other (the
    layout behind the overlapped RCCL exchange): same trajectory as the unsharded sampler."""
    region, X0 = _case()
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, lchain, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=block)
    r = _run_ranks(world, False, tmp_path, parts=parts, block=block)
    assert np.array_equal(r["X"], chain[-1])
    assert np.array_equal(r["lnp"], lchain[-1])
    assert np.array_equal(r["nacc"], nacc)


@pytest.mark.parametrize("world,parts,block", [(2, 1, 8), (2, 2, 4), (4, 2, 4), (8, 1, 4)])
def test_sharded_host_abi_matches_oracle_cpu(world, parts, block, tmp_path):
    """The walker-sharded driver over gloo with the HOST implementation of the C ABI behind ctypes
    (real set_shard_parts / half_step_part / pack_get / scatter_put calls on every rank): the
    trajectory of the unsharded oracle sampler, for every world size and piece count."""
    region, X0 = _case()
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, lchain, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=block)
    r = _run_ranks(world, "cpu_abi", tmp_path, parts=parts, block=block)
    assert np.allclose(r["X"], chain[-1], rtol=1e-10, atol=1e-12) and np.array_equal(r["nacc"], nacc)
    assert np.allclose(r["lnp"], lchain[-1], rtol=1e-9, atol=1e-9)


def _region_batch_case():
    g = load_golden("lnprob_cases.npz")
    names = [f"H1215_r{i}_K{k}_m1_sd0" for i, k in ((0, 1), (1, 4), (2, 1), (0, 4), (2, 4))]
    xs, fs, ns = [g[n + "_x"] for n in names], [g[n + "_flux"] for n in names], [g[n + "_noise"] for n in names]
    ks = [1, 4, 1, 4, 4]
    rng = np.random.default_rng(12)
    th = []
    for x, k in zip(xs, ks):
        t = np.empty((16, 4 * k))
        for j in range(k):
            t[:, 4 * j:4 * j + 4] = np.stack([rng.uniform(0.3, 1.5, 16), rng.uniform(x[2], x[-3], 16), rng.uniform(0.5, 3, 16),
                                              rng.uniform(2, 8, 16)], 1)
        th.append(t)
    return xs, fs, ns, ks, th


def _region_worker(rank, world, port, kind, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import vamp_amd
    from vamp_amd.ensemble import RegionShardedBatch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs, fs, ns, ks, th = _region_batch_case()
    lib = vamp_amd._lib.bind(os.path.join(ROOT, "oracle", "libvamp_cpu.so")) if kind == "cpu_abi" else None
    ctx = vamp_amd.HipContext(device=0, lib=lib)
    batch = RegionShardedBatch(ctx, xs, fs, ns, ks, th, seed=31337, mode=vamp_amd.MODE_VOIGT4, dist=dist, split_block=8)
    assert sorted(sum(batch.assignment, [])) == list(range(5)) and batch.mine == batch.assignment[rank]
    mine, _ = batch.run(5)
    allres = batch.gather(mine)
    if rank == 0:
        np.savez(os.path.join(out_dir, "regions.npz"), **{f"chain_{r}": v["chain"] for r, v in allres.items()},
                 **{f"nacc_{r}": v["n_accept"] for r, v in allres.items()})
    dist.barrier()
    dist.destroy_process_group()


def _check_region_batch(kind, tmp_path):
    import torch.multiprocessing as mp
    import vamp_amd
    mp.spawn(_region_worker, args=(2, _free_port(), kind, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "regions.npz"))
    xs, fs, ns, ks, th = _region_batch_case()
    lib = vamp_amd._lib.bind(os.path.join(ROOT, "oracle", "libvamp_cpu.so")) if kind == "cpu_abi" else None
    with vamp_amd.HipContext(device=0, lib=lib) as ctx:           # the single-context batch of all five regions
        ctx.set_regions(xs, fs, ns, ks, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(th, seed=31337, split_block=8)
        ref = ctx.run(5)
    for r in range(5):
        assert np.array_equal(got[f"chain_{r}"], ref["chain"][r]), r          # same library, same draws: bit for bit
        assert np.array_equal(got[f"nacc_{r}"], ref["n_accept"][r]), r
        reg = vo.Region(x=xs[r], flux=fs[r], noise=ns[r], n_comp=ks[r], mode=vo.MODE_VOIGT4)
        fn = lambda q, reg=reg: vo.log_prob_batch(reg, q)
        chain, _, nacc = vo.run_sampler(fn, th[r], fn(th[r]), 5, seed=31337, block=8, region=r, walker_off=r * 16)
        assert np.allclose(got[f"chain_{r}"], chain, rtol=1e-10, atol=1e-12) and np.array_equal(got[f"nacc_{r}"], nacc), r


def test_region_sharded_batch_matches_single_context_cpu(tmp_path):
    """BASELINE.json config 3 on several devices: independent regions spread over 2 ranks (balanced by
    W * P * K, no collective), each rank its own context of the host ABI.  Every region's chain equals
    the one it has in the single-context batch of all regions, and the oracle's."""
    _check_region_batch("cpu_abi", tmp_path)


@pytest.mark.gpu
def test_region_sharded_batch_matches_single_context_gpu(tmp_path):
    """the same with the HIP kernels (both ranks on device 0)"""
    _check_region_batch("hip", tmp_path)


def test_shard_regions_balances_and_is_deterministic():
    from vamp_amd.ensemble import shard_regions
    costs = [5, 1, 1, 1, 9, 3, 3, 2]
    for world in (1, 2, 3, 8):
        parts = shard_regions(costs, world)
        assert sorted(sum(parts, [])) == list(range(8)) and parts == shard_regions(costs, world)
        loads = [sum(costs[r] for r in p) for p in parts]
        assert max(loads) <= max(max(costs), -(-sum(costs) // world) + max(costs) // 2)


def test_single_rank_pieces_match_run_sampler():
    from vamp_amd.ensemble import ShardedEnsemble
    region, X0 = _case()
    ens = ShardedEnsemble(OracleBackend(region), X0, seed=4242, split_block=8, exchange="none", parts=4)
    assert ens.own_count == X0.shape[0] and len(ens.own_ranges) == 4
    ens.step(6)
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, _, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=8)
    X, lnp, na = ens.gather_state()
    assert np.array_equal(X, chain[-1]) and np.array_equal(na, nacc)


def test_single_rank_driver_matches_run_sampler():
    from vamp_amd.ensemble import ShardedEnsemble
    region, X0 = _case()
    ens = ShardedEnsemble(OracleBackend(region), X0, seed=4242, split_block=8, exchange="none")
    ens.step(6)
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, _, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=8)
    X, lnp, na = ens.gather_state()
    assert np.array_equal(X, chain[-1]) and np.array_equal(na, nacc)


def test_shard_needs_whole_chunks():
    from vamp_amd.ensemble import ShardedEnsemble

    class FakeDist:
        def get_rank(self): return 0
        def get_world_size(self): return 3

    region, X0 = _case()
    with pytest.raises(ValueError):
        ShardedEnsemble(OracleBackend(region), X0, seed=1, split_block=8, dist=FakeDist(), exchange="gloo_host")


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_real_kernels(tmp_path):
    """HIP kernels under 2-way walker sharding, each share in two pieces (both ranks on device 0,
    host-staged gloo exchange), reproduce the single-rank HIP trajectory and the oracle's."""
    import vamp_amd
    region, X0 = _case()
    r = _run_ranks(2, True, tmp_path, parts=2)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=4242, split_block=8)
        ctx.run(6, store_chain=False)
        X1, lnp1, nacc1, _ = ctx.get_state()
    assert np.array_equal(r["X"], X1) and np.array_equal(r["nacc"], nacc1)
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, _, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=8)
    assert np.allclose(r["X"], chain[-1], rtol=1e-10, atol=1e-12) and np.array_equal(r["nacc"], nacc)


def _big_case(W=512):
    region, _ = _case()
    rng = np.random.default_rng(18)
    X0 = np.stack([rng.uniform(0.3, 1.5, W), rng.uniform(-4, 4, W), rng.uniform(0.5, 3, W), rng.uniform(2, 8, W)], 1)
    return region, X0


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 2, 4])
def test_in_library_rccl_exchange_single_rank(parts):
    """The production exchange path on one GPU: an RCCL communicator of one rank inside the library
    (vamp_comm_init_rank), movers packed by the half-step kernel, ncclAllGather + scatter on the
    library's streams (communication stream when the share is cut into pieces), the whole loop in
    one vamp_sampler_run_dev call.  Same trajectory as the plain sampler, bit for bit, and the packed
    rows are exactly the rows the movers ended with."""
    import vamp_amd
    from vamp_amd.ensemble import ShardedEnsemble
    region, X0 = _big_case()
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=99, split_block=32)
        ref = ctx.run(8)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ens = ShardedEnsemble(ctx, X0, seed=99, split_block=32, exchange="rccl", exchange_single_rank=True, parts=parts)
        assert ens.parts == parts and ens.own_count == X0.shape[0]
        ens.step(3)                               # half-step by half-step from the host
        X3, lnp3, _, step = ctx.get_state()
        assert step == 3 and np.array_equal(X3, ref["chain"][2]) and np.array_equal(lnp3, ref["lnprob"][2])
        # the movers of the last (blue) half-step, as packed for the wire
        red, blue = vo.split_tables(99, 2, X0.shape[0], 32)
        n = X0.shape[0] // 2 // parts
        for p in range(parts):
            rows = ctx.pack_get(p)
            ws = blue[p * n:(p + 1) * n]
            assert np.array_equal(rows[:, :-1], X3[ws]) and np.array_equal(rows[:, -1], lnp3[ws])
        res = ctx.run(5)                          # the same through vamp_sampler_run
        assert np.array_equal(res["chain"], ref["chain"][3:]) and np.array_equal(res["n_accept"], ref["n_accept"])


@pytest.mark.gpu
def test_scatter_rewrites_foreign_rows_only():
    """vamp_sampler_scatter_put on a 2-shard context: rows of the other rank's movers are written
    where this (step, half) puts those slots, own rows and the frozen colour are left alone."""
    import vamp_amd
    region, X0 = _big_case(64)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=5, split_block=16)
        (b, e), = ctx.sampler_set_shard_parts(1, 2, 1)            # this ctx plays rank 1 of 2
        assert (b, e) == (32, 64)
        ctx.half_step_part(0, 0)
        mine = ctx.pack_get(0)
        X1, lnp1, _, _ = ctx.get_state()
        red, blue = vo.split_tables(5, 0, 64, 16)
        assert np.array_equal(mine[:, :-1], X1[red[16:]]) and np.array_equal(mine[:, -1], lnp1[red[16:]])
        foreign = np.arange(16 * 5, dtype=np.float64).reshape(16, 5) + 1000.0
        ctx.scatter_put(0, np.concatenate([foreign, -mine]))          # own block deliberately wrong: must be ignored
        X2, lnp2, _, _ = ctx.get_state()
        assert np.array_equal(X2[red[:16]], foreign[:, :-1]) and np.array_equal(lnp2[red[:16]], foreign[:, -1])
        keep = np.ones(64, dtype=bool)
        keep[red[:16]] = False
        assert np.array_equal(X2[keep], X1[keep]) and np.array_equal(lnp2[keep], lnp1[keep])


@pytest.mark.gpu
def test_default_stream_orders_with_torch_and_run_dev_records_on_device():
    """(1) vamp_ctx_set_stream_default: half-steps issued on HIP's legacy default stream are ordered
    with torch work on torch's default stream -- snapshots of the bound state taken from the torch
    side after every step, with no host synchronisation in between, equal the chain of a plain run.
    (2) vamp_sampler_run_dev writes the chain into caller-owned device memory."""
    import torch
    import vamp_amd
    region, X0 = _big_case(2048)
    dev = torch.device("cuda", 0)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=31, split_block=64)
        ref = ctx.run(6)
        # (2) device-resident chain
        ctx.sampler_init(X0, seed=31, split_block=64)
        chain_t = torch.zeros((3, X0.size), dtype=torch.float64, device=dev)
        lnp_t = torch.zeros((3, X0.shape[0]), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.run_dev(6, thin=2, chain_ptr=chain_t.data_ptr(), lnprob_ptr=lnp_t.data_ptr())
        assert np.array_equal(chain_t.cpu().numpy().reshape(3, *X0.shape), ref["chain"][1::2])
        assert np.array_equal(lnp_t.cpu().numpy(), ref["lnprob"][1::2])
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        X_t = torch.empty(X0.size, dtype=torch.float64, device=dev)
        L_t = torch.empty(X0.shape[0], dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.set_stream_default()
        ctx.sampler_bind_state(X_t.data_ptr(), L_t.data_ptr())
        ctx.sampler_init(X0, seed=31, split_block=64)
        snaps = []
        for _ in range(6):
            ctx.half_step(0)
            ctx.half_step(1)
            snaps.append(X_t.clone())             # torch's default stream: ordered after the kernels
        torch.cuda.synchronize()
        got = np.stack([s.cpu().numpy().reshape(X0.shape) for s in snaps])
        assert np.array_equal(got, ref["chain"])


def _fake_rccl_worker(rank, world, port, out_dir, parts):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      VAMP_RCCL_LIB=os.path.join(ROOT, "tests", "host", "librccl_fake.so"))
    import torch.distributed as dist
    import vamp_amd
    from vamp_amd.ensemble import ShardedEnsemble
    dist.init_process_group("gloo", rank=rank, world_size=world)
    region, X0 = _big_case(256)
    ctx = vamp_amd.HipContext(device=0)
    ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
    ens = ShardedEnsemble(ctx, X0, seed=77, split_block=16, dist=dist, exchange="rccl", parts=parts)
    assert ens.exchange == "rccl" and ens.own_count == X0.shape[0] // world
    ens.step(2)                                   # half-step by half-step ...
    ens.run_dev(4)                                # ... and the whole loop in one library call
    X, lnp, nacc = ens.gather_state()
    if rank == 0:
        np.savez(os.path.join(out_dir, "fake.npz"), X=X, lnp=lnp, nacc=nacc)
    dist.barrier()
    ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,parts", [(2, 1), (2, 2), (4, 2)])
def test_in_library_exchange_several_ranks_on_one_gpu(world, parts, tmp_path):
    """The production multi-GPU path end to end at world > 1 on the one-GPU box: vamp_comm_unique_id
    -> id over gloo -> vamp_comm_init_rank -> set_shard_parts -> pack in the half-step kernel ->
    all-gather -> k_scatter_rows, driven by vamp_sampler_half_step and vamp_sampler_run_dev.  The wire
    is a stand-in for RCCL (tests/host/rccl_fake.cpp, host shared memory; RCCL refuses two ranks on
    one device); everything else is the code the 8-GPU run executes.  Same chain as one rank."""
    import torch.multiprocessing as mp
    import vamp_amd
    if not os.path.exists(os.path.join(ROOT, "tests", "host", "librccl_fake.so")):
        import __graft_entry__ as ge
        ge.build()
    mp.spawn(_fake_rccl_worker, args=(world, _free_port(), str(tmp_path), parts), nprocs=world, join=True)
    r = np.load(os.path.join(str(tmp_path), "fake.npz"))
    region, X0 = _big_case(256)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=77, split_block=16)
        ctx.run(6, store_chain=False)
        X1, lnp1, nacc1, _ = ctx.get_state()
    assert np.array_equal(r["X"], X1) and np.array_equal(r["lnp"], lnp1) and np.array_equal(r["nacc"], nacc1)


@pytest.mark.gpu
def test_pieces_of_a_packed_single_region_ensemble():
    """A short one-line region with 65 536 walkers runs four walkers per wavefront with draws from
    k_draws; cut into pieces (the layout of the overlapped exchange) and with the in-library
    exchange of one rank, the chain is the unsharded one: the draw and pack indices are local to a
    piece's launch, the slots global."""
    import vamp_amd
    from vamp_amd.ensemble import ShardedEnsemble
    region, _ = _case()
    rng = np.random.default_rng(21)
    W = 65536
    X0 = np.stack([rng.uniform(0.3, 1.5, W), rng.uniform(-4, 4, W), rng.uniform(0.5, 3, W), rng.uniform(2, 8, W)], 1)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=8, split_block=1024)
        ctx.run(3, store_chain=False)
        X1, lnp1, nacc1, _ = ctx.get_state()
    for exchange, single in (("none", False), ("rccl", True)):
        with vamp_amd.HipContext(device=0) as ctx:
            ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
            ens = ShardedEnsemble(ctx, X0, seed=8, split_block=1024, exchange=exchange, exchange_single_rank=single, parts=2)
            assert ens.parts == 2
            ens.step(1)
            ens.run_dev(2)
            X2, lnp2, nacc2, step = ctx.get_state()
            assert step == 3 and np.array_equal(X2, X1) and np.array_equal(lnp2, lnp1) and np.array_equal(nacc2, nacc1), exchange


# ---- round 3: bench.py as the driver runs it, fall-back safety net, call order, shape under sharding ----
FAKE_RCCL = os.path.join(ROOT, "tests", "host", "librccl_fake.so")
# (a short second run too: its length is derived from the timed run, and every rank must derive the same one)
BENCH_SMALL = ["--walkers", "4096", "--pixels", "2048", "--components", "6", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--sustain-seconds", "0.2"]


def _run_bench(extra_args, extra_env, timeout=600):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0]), out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4])
def test_bench_plain_form_launches_its_own_ranks(n):
    """`python3 bench.py --gpus N` with no WORLD_SIZE in the environment (the form the driver uses at
    N = 1) starts its N ranks itself -- a GPU-free parent, torch.distributed.run as a child -- and
    prints ONE line that proves how many ranks the exchange spanned.  N ranks share device 0
    (VAMP_BENCH_DEVICE) over the stand-in for RCCL; everything else is the 8-GPU code path."""
    line, err = _run_bench(["--gpus", str(n)] + BENCH_SMALL, {"VAMP_BENCH_DEVICE": "0", "VAMP_RCCL_LIB": FAKE_RCCL})
    assert "starting %d ranks" % n in err
    assert line["n_gpus"] == n and line["rccl_ranks"] == n and line["scaling"] == "strong"
    ex = line["exchange"]
    assert ex["kind"] == "rccl" and ex["rccl_ranks"] == n and ex["rccl_ranks_is"].startswith("ncclCommCount")
    from vamp_amd.hip_backend import default_split_block
    chunks = 4096 // default_split_block(4096, n)                  # pieces must be whole split chunks on every rank
    assert set(ex["parts_tried_ms_per_step"]) == {str(p) for p in (1, 2) if chunks % (n * p) == 0}
    assert str(ex["parts"]) in ex["parts_tried_ms_per_step"]
    assert ex["exchanges_timed"] == 2 * 2 * ex["parts"]             # 2 steps x 2 colours x pieces
    assert ex["kernel_ms_per_half_step"] > 0 and ex["exchange_ms_per_half_step"] > 0 and ex["wall_ms_per_half_step"] > 0
    assert 0.0 <= ex["overlap_frac"] <= 1.0
    assert ex["bytes_per_rank_per_half_step"] == (4096 // n // 2) * (18 + 1) * 8
    assert line["value"] > 0 and 0.05 < line["acceptance_fraction"] < 0.95 and line["finite_lnprob_fraction"] == 1.0
    assert line["roofline"]["walker_steps_per_launch"] == 4096 // n // 2 // ex["parts"]
    assert line["sustained"]["steps"] >= 2 and line["sustained"]["value"] > 0
    # the line proves what carried its bytes and where its ranks ran: this is a rehearsal and says so
    assert line["rehearsal"] is True and line["rehearsal_knobs"] == {"VAMP_BENCH_DEVICE": "0", "VAMP_RCCL_LIB": FAKE_RCCL}
    assert os.path.samefile(ex["rccl_library"], FAKE_RCCL)
    assert len(line["device_of_rank"]) == n and ex["distinct_devices"] == 1
    assert sorted(d["rank"] for d in line["device_of_rank"]) == list(range(n)) and {d["device_index"] for d in line["device_of_rank"]} == {0}


@pytest.mark.gpu
def test_bench_driver_form_is_not_a_rehearsal():
    """The N = 1 form the driver runs: no knob in the environment -> `rehearsal` false, one device named; with
    --force-dist the exchange runs over the REAL RCCL of the ROCm runtime the library is bound to, and the line
    names that file."""
    env = {}
    line, _ = _run_bench(BENCH_SMALL, env)
    assert line["n_gpus"] == 1 and line["rehearsal"] is False and line["rehearsal_knobs"] == {} and line["exchange"] is None
    assert len(line["device_of_rank"]) == 1 and line["device_of_rank"][0]["device_index"] == 0
    line, _ = _run_bench(["--force-dist"] + BENCH_SMALL, env)
    ex = line["exchange"]
    assert line["rehearsal"] is False and line["rccl_ranks"] == 1 and ex["kind"] == "rccl"
    assert "librccl" in os.path.basename(ex["rccl_library"]) and ex["distinct_devices"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("fail", ["id", "one_rank"])
def test_bench_falls_back_together_when_the_communicator_fails(fail):
    """The scaling run's safety net: a communicator that cannot be made -- the id on
    rank 0, or ncclCommInitRank on ONE rank while the other is already inside it -- sends EVERY rank
    to the host-staged gloo exchange; nobody is left in a collective alone, and the line says so."""
    env = {"VAMP_BENCH_DEVICE": "0", "VAMP_RCCL_LIB": FAKE_RCCL, "VAMP_COMM_INIT_TIMEOUT": "15"}
    env.update({"VAMP_FAKE_RCCL_FAIL_ID": "1"} if fail == "id" else {"VAMP_FAKE_RCCL_FAIL_INIT_RANK": "1"})
    line, err = _run_bench(["--gpus", "2"] + BENCH_SMALL, env)
    assert "FALLING BACK" in err
    assert line["n_gpus"] == 2 and line["rccl_ranks"] is None
    assert line["exchange"]["kind"].startswith("gloo_host") and "FALLBACK" in line["config"]["exchange"]
    assert line["value"] > 0 and line["finite_lnprob_fraction"] == 1.0


def _order_worker(rank, world, port, out_dir):
    """shard first, communicator second (the order vamp_hip.h now allows), two pieces"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VAMP_RCCL_LIB=FAKE_RCCL)
    import torch.distributed as dist
    import vamp_amd
    from vamp_amd import _lib
    from vamp_amd.hip_backend import comm_unique_id
    dist.init_process_group("gloo", rank=rank, world_size=world)
    region, X0 = _big_case(256)
    ctx = vamp_amd.HipContext(device=0)
    ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
    ctx.sampler_init(X0, seed=77, split_block=16)
    ctx.sampler_set_shard_parts(rank, world, 2)                  # BEFORE the communicator exists
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    with pytest.raises(_lib.VampError) as ei:                    # a communicator that contradicts the shard
        ctx.comm_init_rank(box[0], (rank + 1) % world, world)
    assert ei.value.code == -1
    ctx.comm_init_rank(box[0], rank, world)
    assert ctx.comm_info() == (rank, world, True)
    ctx.kernel_timing(True)
    ctx.run_dev(6)
    x_ms, x_n = ctx.exchange_timing()
    k_ms, k_n = ctx.kernel_timing(False)
    assert x_n == 6 * 2 * 2 and k_n == 6 * 2 * 2 and x_ms > 0 and k_ms > 0
    X, lnp, _, _ = ctx.get_state()
    if rank == 0:
        np.savez(os.path.join(out_dir, "order.npz"), X=X, lnp=lnp)
    dist.barrier()
    ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_shard_before_communicator_is_a_valid_order(tmp_path):
    """sampler_init -> set_shard_parts(world 2, 2 pieces) -> comm_init_rank used to leave
    the per-piece events uncreated (out-of-bounds in exchange_part).  Now either order works, a
    mismatch of rank/world is VAMP_ERR_ARG, and the chain is the single-rank one."""
    import torch.multiprocessing as mp
    import vamp_amd
    mp.spawn(_order_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), "order.npz"))
    region, X0 = _big_case(256)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=77, split_block=16)
        ctx.run(6, store_chain=False)
        X1, lnp1, _, _ = ctx.get_state()
    assert np.array_equal(r["X"], X1) and np.array_equal(r["lnp"], lnp1)


@pytest.mark.gpu
def test_small_pieces_of_a_packed_ensemble_keep_its_shape():
    """the kernel shape of a launch class used to follow the size of THIS launch, so a
    short-region ensemble of 65 536 walkers ran four walkers per wavefront on one device and one per
    wavefront as a 4096-walker piece (world 4 x 2 pieces) -- agreeing to rounding only.  The shape now
    follows the unsharded ensemble: eight 4096-mover pieces give the unsharded chain bit for bit."""
    import vamp_amd
    region, _ = _case()
    rng = np.random.default_rng(22)
    W = 65536
    X0 = np.stack([rng.uniform(0.3, 1.5, W), rng.uniform(-4, 4, W), rng.uniform(0.5, 3, W), rng.uniform(2, 8, W)], 1)
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=9, split_block=1024)
        ctx.run(2, store_chain=False)
        X1, lnp1, nacc1, _ = ctx.get_state()
    with vamp_amd.HipContext(device=0) as ctx:
        ctx.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
        ctx.sampler_init(X0, seed=9, split_block=1024)
        ctx.sampler_set_shard_parts(0, 1, 8)                      # 8 pieces of 4096 movers: below the packing threshold
        for _ in range(2):
            for half in (0, 1):
                for p in range(8):
                    ctx.half_step_part(half, p)
        X2, lnp2, nacc2, step = ctx.get_state()
    assert step == 2 and np.array_equal(X2, X1) and np.array_equal(lnp2, lnp1) and np.array_equal(nacc2, nacc1)


def _join_fail_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import vamp_amd
    from vamp_amd.ensemble import ShardedEnsemble
    dist.init_process_group("gloo", rank=rank, world_size=world)
    region, X0 = _case()
    backend = vamp_amd.HipContext(lib=vamp_amd._lib.bind(os.path.join(ROOT, "oracle", "libvamp_cpu.so")))
    backend.set_regions(region.x, region.flux, region.noise, 1, mode=vamp_amd.MODE_VOIGT4)
    code = 0
    try:       # the host build refuses a communicator of world > 1: VAMP_ERR_COMM inside comm_init_rank
        ShardedEnsemble(backend, X0, seed=1, split_block=8, dist=dist, exchange="rccl", parts=1)
    except vamp_amd._lib.VampError as e:
        code = e.code
    # every rank is here, in step: a collective right after the failure completes (bench.py's fall-back does this)
    flag = torch.tensor([code])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    ens = ShardedEnsemble(backend, X0, seed=4242, split_block=8, dist=dist, exchange="gloo_host", parts=1)
    ens.step(2)
    X, lnp, nacc = ens.gather_state()
    if rank == 0:
        np.savez(os.path.join(out_dir, "joinfail.npz"), X=X, code=code, flag=int(flag[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_failed_communicator_raises_on_every_rank_cpu(tmp_path):
    """ShardedEnsemble(exchange='rccl') either gives every rank a communicator or raises
    VampError(-3) on every rank TOGETHER (here: the host build of the ABI, which has no RCCL), so the
    caller's next collective -- bench.py's fall-back to the host-staged exchange -- cannot hang."""
    import torch.multiprocessing as mp
    mp.spawn(_join_fail_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = np.load(os.path.join(str(tmp_path), "joinfail.npz"))
    assert int(r["code"]) == -3 and int(r["flag"]) == -3
    region, X0 = _case()
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, _, _ = vo.run_sampler(fn, X0, fn(X0), 2, seed=4242, block=8)
    assert np.allclose(r["X"], chain[-1], rtol=1e-10, atol=1e-12)
