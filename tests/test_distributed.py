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
        return [(2 * b, 2 * e) for b, e in self.part_slots]

    def half_step_part(self, half, part):
        self.slot_begin, self.slot_end = self.part_slots[part]
        step = self.step
        self.half_step(half)
        if half == 1 and part != len(self.part_slots) - 1:
            self.step = step              # the counter advances after the last piece

    def half_step(self, half):
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
    if use_gpu:
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


@pytest.mark.parametrize("world,parts,block", [(2, 2, 8), (4, 2, 4), (2, 4, 4)])
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


def test_single_rank_pieces_match_run_sampler():
    from vamp_amd.ensemble import ShardedEnsemble
    region, X0 = _case()
    ens = ShardedEnsemble(OracleBackend(region), X0, seed=4242, split_block=8, exchange="none", torch_state=False, parts=4)
    assert ens.own_count == X0.shape[0] and len(ens.own_ranges) == 4
    ens.step(6)
    fn = lambda q: vo.log_prob_batch_fast(region, q)
    chain, _, nacc = vo.run_sampler(fn, X0, fn(X0), 6, seed=4242, block=8)
    X, lnp, na = ens.gather_state()
    assert np.array_equal(X, chain[-1]) and np.array_equal(na, nacc)


def test_single_rank_driver_matches_run_sampler():
    from vamp_amd.ensemble import ShardedEnsemble
    region, X0 = _case()
    ens = ShardedEnsemble(OracleBackend(region), X0, seed=4242, split_block=8, exchange="none", torch_state=False)
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
