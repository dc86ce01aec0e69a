"""Walker-parallel stretch-move driver: one process per GPU.

The ensemble of ONE region is sharded over ``world`` ranks by blocks of whole split chunks, so
every rank moves exactly W/(2*world) walkers per half-step.  Every rank keeps the full state
X[W, D] on its device, because a moving walker may pick any member of the frozen complement as
its partner.  After a half-step only the ACTIVE colour has changed: each rank contributes the
(W/2/world) x (D + 1) doubles its movers ended with (position + lnprob, packed in slot order by
the half-step kernel itself), one all-gather moves them, and a scatter kernel writes them to the
rows that hold those slots in this (step, half) -- SURVEY 8e: 1.5 MiB per GPU per half-step on
the headline at world = 8.

``exchange``:
  "rccl"       production: the library owns an RCCL communicator (vamp_comm_init_rank) and runs
               pack -> ncclAllGather -> scatter on its own HIP streams inside
               vamp_sampler_half_step / vamp_sampler_run_dev; nothing here touches the data path
               and there is no host synchronisation inside the step loop.  ``dist`` (a
               torch.distributed process group, any backend) is used ONCE, to hand rank 0's
               128-byte communicator id to the other ranks.
  "gloo_host"  tests: the same pack and scatter kernels, with the packed rows staged through host
               memory and all-gathered by ``dist`` (lets two ranks share one GPU, and lets the
               host logic run against an oracle-backed stand-in on CPU).
  "none"       single rank.

Overlap: a rank's share can be cut into ``parts`` pieces (default 1; ``VAMP_EXCHANGE_PARTS``); the
exchange of piece p runs on the library's communication stream while the kernel of piece p + 1 computes.  Safe because a
half-step kernel reads only rows of the frozen colour and writes only its own piece, while the
scatter of piece p writes only rows of the moving colour of piece p.

Independent regions (BASELINE.json config 3) need no exchange at all: see ``shard_regions``.

Counter-based draws are keyed by (seed, step, half, global walker id) and the red/blue split by
(seed, step, chunk), so the trajectory is bit-identical for every ``world`` and ``parts``
(tests/test_distributed.py).
"""
from __future__ import annotations

import os

import numpy as np


def shard_regions(costs, world):
    """Longest-processing-time assignment of independent regions to ``world`` ranks.
    ``costs[r]`` ~ W_r * P_r * K_r (SURVEY 8e "Multi-region (C3)": no collective, balance by the
    sum).  Returns a list of ``world`` sorted index lists; every rank computes the same answer."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for r in order:
        g = int(np.argmin(load))          # ties: lowest rank
        out[g].append(int(r))
        load[g] += costs[r]
    return [sorted(v) for v in out]


class RegionShardedBatch:
    """The independent regions of one spectrum spread over the ranks of ``dist`` (BASELINE.json
    config 3 on several GPUs; SURVEY 8e "Multi-region"): every rank samples its own regions in its own
    context, there is NO collective in the data path, and ``dist`` is only used to hand results
    round afterwards.  Regions are assigned by ``shard_regions`` on the cost W * P * K; every region
    keeps its index in the whole spectrum as its identity in the draw keys
    (vamp_set_region_ids), so its chain is the one the single-context batch produces.

    ``backend``: a ``HipContext`` on this rank's device (tests: the host implementation of the ABI).
    ``xs / fluxes / noises / n_comp / thetas0``: per-region lists for the WHOLE spectrum, identical on
    every rank."""

    def __init__(self, backend, xs, fluxes, noises, n_comp, thetas0, seed, mode, dist=None, split_block=None, costs=None,
                 **region_kw):
        self.backend, self.dist = backend, dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.n_regions = len(xs)
        W = np.asarray(thetas0[0]).shape[0]
        if costs is None:
            costs = [float(W) * len(x) * int(k) for x, k in zip(xs, n_comp)]
        self.assignment = shard_regions(costs, self.world)
        self.mine = self.assignment[self.rank]
        if not self.mine:
            raise ValueError("more ranks than regions")
        pick = lambda seq: [seq[r] for r in self.mine]
        backend.set_regions(pick(xs), pick(fluxes), pick(noises), pick(list(n_comp)), mode=mode, **region_kw)
        backend.set_region_ids(self.mine)
        if split_block is None:
            from .hip_backend import default_split_block
            split_block = default_split_block(W)
        backend.sampler_init([np.ascontiguousarray(thetas0[r], dtype=np.float64) for r in self.mine], seed=seed,
                             split_block=split_block)

    def run(self, n_steps, thin=1, store_chain=True):
        """This rank's regions: {global region index: dict(chain, lnprob, n_accept)} and the seconds."""
        res = self.backend.run(n_steps, thin=thin, store_chain=store_chain)
        one = len(self.mine) == 1
        out = {}
        for i, r in enumerate(self.mine):
            out[r] = {"n_accept": res["n_accept"] if one else res["n_accept"][i]}
            if store_chain:
                out[r]["chain"] = res["chain"] if one else res["chain"][i]
                out[r]["lnprob"] = res["lnprob"] if one else res["lnprob"][i]
        return out, res["seconds"]

    def gather(self, mine):
        """Results of every rank on every rank (host side, after the run)."""
        if self.world == 1:
            return dict(mine)
        box = [None] * self.world
        self.dist.all_gather_object(box, mine)
        out = {}
        for part in box:
            out.update(part)
        return out


class ShardedEnsemble:
    """Drive a single-region sampler whose walkers are sharded over the ranks of ``dist``.

    ``backend`` is a ``HipContext`` (or, in the CPU tests of the host logic, any object with the
    same ``sampler_*`` / ``half_step_part`` / ``pack_get`` / ``scatter_put`` surface)."""

    def __init__(self, backend, theta0, seed, a=2.0, split_block=None, dist=None, exchange="rccl", parts=None,
                 exchange_single_rank=False):
        self.backend = backend
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        # exchange_single_rank: keep the collective in the loop even for one rank (rehearses the
        # RCCL call pattern on a one-GPU box; the gather is then a self-copy)
        self.exchange = exchange if (self.world > 1 or exchange_single_rank) else "none"
        if self.exchange not in ("rccl", "gloo_host", "none"):
            raise ValueError("exchange must be 'rccl', 'gloo_host' or 'none'")
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        self.W, self.D = theta0.shape
        if split_block is None:
            from .hip_backend import default_split_block
            split_block = default_split_block(self.W, self.world)
        if (self.W // split_block) % self.world:
            raise ValueError("W/split_block must be a multiple of the number of ranks")
        self.split_block = split_block
        if parts is None:
            # one piece until a multi-GPU run shows that the overlap pays: two 2048-walker pieces cost 8 % more kernel
            # time than one 4096-walker launch at world 8 (profiles/r02_g_headline_shard_sizes.txt); bench.py times both
            parts = int(os.environ.get("VAMP_EXCHANGE_PARTS", "0")) or 1
            while parts > 1 and (self.W // split_block) % (self.world * parts):
                parts -= 1
        if parts < 1 or (self.W // split_block) % (self.world * parts):
            raise ValueError("W/split_block must be a multiple of ranks * parts")
        self.parts = parts
        if self.exchange == "rccl":
            self._join_communicator(backend, dist)
        backend.sampler_init(theta0, seed=seed, a=a, split_block=split_block)
        if self.world > 1 or self.parts > 1 or self.exchange == "rccl":
            self.own_ranges = backend.sampler_set_shard_parts(self.rank, self.world, self.parts)
        else:
            self.own_ranges = [(0, self.W)]
        self.own_begin, self.own_end = self.own_ranges[0][0], self.own_ranges[-1][1]   # contiguous iff parts == 1
        self.own_count = sum(e - b for b, e in self.own_ranges)
        self.own_mask = np.zeros(self.W, dtype=bool)
        for b, e in self.own_ranges:
            self.own_mask[b:e] = True
        self.steps_done = 0

    def _join_communicator(self, backend, dist):
        """Every rank of ``dist`` ends this call with a communicator, or EVERY rank raises the same
        VampError(-3): no rank is ever left alone inside a collective (bench.py's fall-back to the
        host-staged exchange relies on that).
          1. every rank checks locally that the RCCL entry points load and an id can be made; rank 0's
             id and every rank's outcome travel over ``dist`` before anyone enters RCCL;
          2. ncclCommInitRank blocks until all ranks have joined, so a rank that fails in it would
             leave the others inside: the call runs in a thread that is given
             ``VAMP_COMM_INIT_TIMEOUT`` seconds (default 300), and the outcomes are exchanged again.
             A rank whose call never returned abandons its context (``backend.abandon()``) instead
             of destroying it under the stuck call."""
        import threading
        from ._lib import VampError
        from .hip_backend import comm_unique_id
        have = getattr(backend, "comm", None)
        if have is not None:                      # a re-built ensemble on a context that already joined
            if have != (self.rank, self.world):
                raise ValueError("the context's communicator has another rank/world")
            return
        err = None
        try:
            my_id = comm_unique_id(getattr(backend, "_lib", None))
        except VampError as e:
            my_id, err = None, str(e)
        if self.world > 1:
            box = [my_id if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)            # 128 bytes over the bootstrap group
            my_id = box[0]
            errs = [None] * self.world
            dist.all_gather_object(errs, err)
            bad = [(r, e) for r, e in enumerate(errs) if e]
            if bad:
                raise VampError(-3, "no RCCL communicator: rank %d: %s" % bad[0])
        elif err:
            raise VampError(-3, err)
        out = {}

        def join():
            try:
                backend.comm_init_rank(my_id, self.rank, self.world)
            except Exception as e:                            # reported below, on every rank
                out["err"] = str(e)

        if self.world == 1:
            join()
            if "err" in out:
                raise VampError(-3, out["err"])
            return
        timeout = float(os.environ.get("VAMP_COMM_INIT_TIMEOUT", "300"))
        t = threading.Thread(target=join, daemon=True)
        t.start()
        t.join(timeout)
        stuck = t.is_alive()
        err = ("vamp_comm_init_rank did not return within %.0f s" % timeout) if stuck else out.get("err")
        errs = [None] * self.world
        dist.all_gather_object(errs, err)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        if bad:
            if stuck and hasattr(backend, "abandon"):
                backend.abandon()
            e = VampError(-3, "no RCCL communicator: rank %d: %s" % bad[0])
            e.stuck = stuck
            raise e

    def _exchange_host(self, p):
        """piece p: packed movers of every rank through host memory and ``dist``"""
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(self.backend.pack_get(p)))
        full = torch.empty((self.world * mine.shape[0], mine.shape[1]), dtype=torch.float64)
        self.dist.all_gather_into_tensor(full, mine)
        self.backend.scatter_put(p, full.numpy())

    def step(self, n_steps=1):
        for _ in range(n_steps):
            for half in (0, 1):
                if self.exchange == "gloo_host":
                    for p in range(self.parts):
                        self.backend.half_step_part(half, p)
                        self._exchange_host(p)
                else:                                       # "rccl": the library exchanges; "none": nothing to do
                    self.backend.half_step(half)
            self.steps_done += 1

    def run_dev(self, n_steps, thin=1, chain_ptr=None, lnprob_ptr=None):
        """The whole loop in one library call (production path; chain kept on the device)."""
        if self.exchange == "gloo_host":
            raise ValueError("the host-staged exchange is stepped from Python (step())")
        sec = self.backend.run_dev(n_steps, thin=thin, chain_ptr=chain_ptr, lnprob_ptr=lnprob_ptr)
        self.steps_done += n_steps
        return sec

    def synchronize(self):
        if hasattr(self.backend, "synchronize"):
            self.backend.synchronize()

    def gather_state(self):
        """Full (X[W,D], lnp[W], n_accept[W]) on every rank.  Positions and lnprob are complete on
        every rank already (they travel with the exchange); n_accept is kept by the owner only and
        is gathered here, outside the step loop."""
        X, lnp, nacc, _ = self.backend.get_state()
        if self.world == 1:
            return X, lnp, nacc
        import torch
        full_np = np.empty(self.W, dtype=np.int64)
        slab = self.W // self.parts
        for p, (b, e) in enumerate(self.own_ranges):
            mine = torch.from_numpy(np.ascontiguousarray(nacc[b:e]))
            full = torch.empty(slab, dtype=torch.int64)
            self.dist.all_gather_into_tensor(full, mine)          # needs a CPU-capable (gloo) group
            full_np[p * slab:(p + 1) * slab] = full.numpy()
        return X, lnp, full_np
