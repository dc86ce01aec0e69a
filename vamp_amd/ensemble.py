"""Walker-parallel stretch-move driver: one process per GPU, torch.distributed for the exchange.

The ensemble of ONE region is sharded over ``world`` ranks by contiguous blocks of walkers (whole
split chunks, so every rank moves exactly W/(2*world) walkers per half-step).  Every rank keeps
the full position array X[W, D] on its device, because a moving walker may pick any member of
the frozen complement as its partner.  After each half-step the ranks exchange the rows they own
with ONE all-gather (RCCL over xGMI when the backend is "nccl"), issued on the same HIP stream
as the kernels so that nothing synchronises with the host inside the step loop.

Independent regions (BASELINE.json config 3) need no exchange at all: give each rank its own
``HipContext`` with a subset of the regions.

Counter-based draws are keyed by (seed, step, half, global walker id) and the red/blue split by
(seed, step, chunk), so the trajectory is bit-identical for every ``world``
(tests/test_distributed.py).
"""
from __future__ import annotations

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
                 torch_state=None, exchange_single_rank=False):
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
        self.own_begin, self.own_end = backend.sampler_set_shard(self.rank, self.world) if self.world > 1 else (0, self.W)
        if torch_state:
            self._X2d = self._X_t.view(self.W, self.D)
            self._own = self._X2d[self.own_begin:self.own_end]
        self.steps_done = 0

    # one exchange of the rows this rank owns
    def _all_gather(self):
        if self.exchange == "none":
            return
        if self.exchange == "nccl":
            self.dist.all_gather_into_tensor(self._X2d, self._own)
            return
        import torch
        X, lnp, nacc, step = self.backend.get_state()
        mine = torch.from_numpy(np.ascontiguousarray(X[self.own_begin:self.own_end]))
        full = torch.empty((self.W, self.D), dtype=torch.float64)
        self.dist.all_gather_into_tensor(full, mine)
        # lnprob of foreign walkers is never read by this rank; keep the local values
        self.backend.set_state(full.numpy(), lnp, step)

    def step(self, n_steps=1):
        for _ in range(n_steps):
            for half in (0, 1):
                self.backend.half_step(half)
                self._all_gather()
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
        for arr, dt in ((lnp, torch.float64), (nacc, torch.int64)):
            mine = torch.from_numpy(np.ascontiguousarray(arr[self.own_begin:self.own_end]))
            if self.exchange == "nccl":
                mine = mine.to(self._X_t.device)
                full = torch.empty(self.W, dtype=dt, device=self._X_t.device)
            else:
                full = torch.empty(self.W, dtype=dt)
            self.dist.all_gather_into_tensor(full, mine)
            outs.append(full.cpu().numpy())
        return X, outs[0], outs[1]
