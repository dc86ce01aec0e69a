"""Walker-parallel stretch-move driver: one process per GPU, torch.distributed for the exchange.

The ensemble of ONE region is sharded over ``world`` ranks by contiguous blocks of walkers (whole
split chunks, so every rank moves exactly W/(2*world) walkers per half-step).  Every rank keeps
the full position array X[W, D] on its device, because a moving walker may pick any member of
the frozen complement as its partner.  After each half-step the ranks exchange the rows they own
by all-gather (RCCL over xGMI when the backend is "nccl") with no host synchronisation inside
the step loop.

Overlap: a rank's share is cut into ``parts`` pieces (default 2 with RCCL).  The kernel of piece
p + 1 runs while the all-gather of piece p is in flight on RCCL's own stream (``async_op``; the
compute stream waits for all pieces before the next colour starts).  That is safe because a
half-step kernel reads only rows of the frozen colour, which the in-flight gather rewrites with
the values they already hold, and writes only its own piece.  Rows are laid out so that piece p
of every rank lies in the p-th ``W/parts`` slab of X: each gather is in place into one contiguous
slab (vamp_sampler_set_shard_parts).

Independent regions (BASELINE.json config 3) need no exchange at all: give each rank its own
``HipContext`` with a subset of the regions.

Counter-based draws are keyed by (seed, step, half, global walker id) and the red/blue split by
(seed, step, chunk), so the trajectory is bit-identical for every ``world``
(tests/test_distributed.py).
"""
from __future__ import annotations

import os

import numpy as np


class ShardedEnsemble:
    """Drive a single-region sampler whose walkers are sharded over ``dist`` ranks.

    ``backend`` is a ``HipContext`` (or, in the CPU tests of the host logic, any object with the
    same ``sampler_*``/``half_step``/``get_state`` surface).  ``exchange``:
      "nccl"       in-place all_gather_into_tensor on device memory (production)
      "gloo_host"  stage the owned rows through host memory and a gloo all-gather (lets two
                   ranks share one GPU in tests; also the CPU-only rehearsal path)
    """

    def __init__(self, backend, theta0, seed, a=2.0, split_block=None, dist=None, exchange="nccl", torch_device=None,
                 torch_state=None, exchange_single_rank=False, parts=None):
        self.backend = backend
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        # exchange_single_rank: keep the collective in the loop even for one rank (rehearses the
        # RCCL call pattern on a one-GPU box; the gather is then a self-copy)
        self.exchange = exchange if (self.world > 1 or (exchange_single_rank and dist is not None)) else "none"
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        self.W, self.D = theta0.shape
        if split_block is None:
            from .hip_backend import default_split_block
            split_block = default_split_block(self.W, self.world)
        if (self.W // split_block) % self.world:
            raise ValueError("W/split_block must be a multiple of the number of ranks")
        self.split_block = split_block
        if parts is None:
            parts = int(os.environ.get("VAMP_EXCHANGE_PARTS", "0")) or (2 if self.exchange == "nccl" else 1)
            while parts > 1 and (self.W // split_block) % (self.world * parts):
                parts -= 1
        if parts < 1 or (self.W // split_block) % (self.world * parts):
            raise ValueError("W/split_block must be a multiple of ranks * parts")
        self.parts = parts
        self._torch = None
        self._X_t = None
        if torch_state is None:
            torch_state = self.exchange == "nccl"
        if self.exchange == "nccl" and not torch_state:
            raise ValueError("the nccl exchange needs the state in torch tensors")
        if torch_state:
            import torch
            self._torch = torch
            dev = torch_device if torch_device is not None else torch.device("cuda", torch.cuda.current_device())
            # state lives in torch tensors so that RCCL can address it; the library adopts the
            # pointers and torch's current stream
            self._X_t = torch.empty(self.W * self.D, dtype=torch.float64, device=dev)
            self._lnp_t = torch.empty(self.W, dtype=torch.float64, device=dev)
            backend.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            backend.sampler_bind_state(self._X_t.data_ptr(), self._lnp_t.data_ptr())
        backend.sampler_init(theta0, seed=seed, a=a, split_block=split_block)
        if self.world > 1 or self.parts > 1:
            self.own_ranges = backend.sampler_set_shard_parts(self.rank, self.world, self.parts)
        else:
            self.own_ranges = [(0, self.W)]
        self.own_begin, self.own_end = self.own_ranges[0][0], self.own_ranges[-1][1]   # contiguous iff parts == 1
        self.own_count = sum(e - b for b, e in self.own_ranges)
        self.own_mask = np.zeros(self.W, dtype=bool)
        for b, e in self.own_ranges:
            self.own_mask[b:e] = True
        if torch_state:
            self._X2d = self._X_t.view(self.W, self.D)
            slab = self.W // self.parts
            self._slabs = [self._X2d[p * slab:(p + 1) * slab] for p in range(self.parts)]
            self._own = [self._X2d[b:e] for b, e in self.own_ranges]
        self.steps_done = 0

    # exchange of piece p of the rows this rank owns; returns a work handle or None
    def _all_gather(self, p):
        if self.exchange == "none":
            return None
        if self.exchange == "nccl":
            return self.dist.all_gather_into_tensor(self._slabs[p], self._own[p], async_op=True)
        import torch
        X, lnp, nacc, step = self.backend.get_state()
        b, e = self.own_ranges[p]
        slab = self.W // self.parts
        mine = torch.from_numpy(np.ascontiguousarray(X[b:e]))
        full = torch.empty((slab, self.D), dtype=torch.float64)
        self.dist.all_gather_into_tensor(full, mine)
        X[p * slab:(p + 1) * slab] = full.numpy()
        # lnprob of foreign walkers is never read by this rank; keep the local values
        self.backend.set_state(X, lnp, step)
        return None

    def step(self, n_steps=1):
        for _ in range(n_steps):
            for half in (0, 1):
                pending = []
                for p in range(self.parts):
                    if self.parts == 1:
                        self.backend.half_step(half)
                    else:
                        self.backend.half_step_part(half, p)
                    w = self._all_gather(p)
                    if w is not None:
                        pending.append(w)
                for w in pending:
                    w.wait()          # the compute stream waits; the host does not
            self.steps_done += 1

    def synchronize(self):
        if self._torch is not None:
            self._torch.cuda.synchronize()
        elif hasattr(self.backend, "synchronize"):
            self.backend.synchronize()

    def gather_state(self):
        """Full (X[W,D], lnp[W], n_accept[W]) on every rank (lnp / n_accept are only valid on
        their owner, so they are exchanged here, outside the step loop)."""
        X, lnp, nacc, _ = self.backend.get_state()
        if self.world == 1:
            return X, lnp, nacc
        import torch
        outs = []
        slab = self.W // self.parts
        for arr, dt in ((lnp, torch.float64), (nacc, torch.int64)):
            full_np = np.empty(self.W, dtype=arr.dtype)
            for p, (b, e) in enumerate(self.own_ranges):
                mine = torch.from_numpy(np.ascontiguousarray(arr[b:e]))
                if self.exchange == "nccl":
                    mine = mine.to(self._X_t.device)
                    full = torch.empty(slab, dtype=dt, device=self._X_t.device)
                else:
                    full = torch.empty(slab, dtype=dt)
                self.dist.all_gather_into_tensor(full, mine)
                full_np[p * slab:(p + 1) * slab] = full.cpu().numpy()
            outs.append(full_np)
        return X, outs[0], outs[1]
