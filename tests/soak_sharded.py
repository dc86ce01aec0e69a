#!/usr/bin/env python3
"""Developer soak (not collected by pytest): the in-library exchange at world 2..4 on ONE GPU over the stand-in for the
wire (tests/host/librccl_fake.so), for random ensemble sizes, split blocks, piece counts, parameterisations and region
lengths (short packed regions and tile-code regions): every configuration must reproduce the single-rank chain bit for
bit.  usage (GPU box): python tests/soak_sharded.py [n_cases]"""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FAKE = os.path.join(ROOT, "tests", "host", "librccl_fake.so")


def case(i):
    rng = np.random.default_rng(31000 + i)
    world = int(rng.choice([2, 3, 4]))
    parts = int(rng.choice([1, 2, 3]))
    block = int(rng.choice([4, 8, 16, 32]))
    chunks = world * parts * int(rng.integers(1, 4))
    W = block * chunks
    P = int(rng.choice([30, 44, 200, 600, 2304]))
    K = int(rng.integers(1, 5)) if P < 2000 else int(rng.integers(3, 9))
    x = np.arange(P, dtype=np.float64) - (P - 1) / 2.0
    span = x[-1] - x[0]
    th = np.empty((W, K, 4))
    th[:, :, 0] = rng.uniform(0.2, 2.0, (W, K))
    th[:, :, 1] = rng.uniform(x[0], x[-1], (W, K))
    th[:, :, 2] = 10.0 ** rng.uniform(-3, np.log10(0.2 * span), (W, K))
    th[:, :, 3] = 10.0 ** rng.uniform(-0.5, np.log10(0.3 * span), (W, K))
    flux = np.clip(1.0 + rng.normal(0, 0.03, P), 0, None)
    return dict(world=world, parts=parts, block=block, W=W, P=P, K=K, x=x, flux=flux, noise=np.full(P, 0.03),
                th=np.ascontiguousarray(th.reshape(W, 4 * K)), seed=500 + i, big=bool(W >= 64 and rng.random() < 0.3))


def worker(rank, world, port, i, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), VAMP_RCCL_LIB=FAKE)
    import torch.distributed as dist
    import vamp_amd
    from vamp_amd.ensemble import ShardedEnsemble
    c = case(i)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = vamp_amd.HipContext(device=0)
    if c["big"]:
        ctx.set_packing(16 if c["K"] <= 8 and c["P"] < 2000 else 0)       # a packed shape under sharding
    ctx.set_regions(c["x"], c["flux"], c["noise"], c["K"], mode=vamp_amd.MODE_VOIGT4)
    ens = ShardedEnsemble(ctx, c["th"], seed=c["seed"], split_block=c["block"], dist=dist, exchange="rccl", parts=c["parts"])
    ens.step(1)
    ens.run_dev(3)
    X, lnp, nacc = ens.gather_state()
    if rank == 0:
        np.savez(os.path.join(out_dir, "case%d.npz" % i), X=X, lnp=lnp, nacc=nacc)
    dist.barrier()
    ctx.comm_destroy()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile
    import torch.multiprocessing as mp
    import vamp_amd
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out = tempfile.mkdtemp()
    bad = 0
    for i in range(n):
        c = case(i)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(worker, args=(c["world"], port, i, out), nprocs=c["world"], join=True)
        r = np.load(os.path.join(out, "case%d.npz" % i))
        with vamp_amd.HipContext(device=0) as ctx:
            if c["big"]:
                ctx.set_packing(16 if c["K"] <= 8 and c["P"] < 2000 else 0)
            ctx.set_regions(c["x"], c["flux"], c["noise"], c["K"], mode=vamp_amd.MODE_VOIGT4)
            ctx.sampler_init(c["th"], seed=c["seed"], split_block=c["block"])
            ctx.run(4, store_chain=False)
            X1, lnp1, nacc1, _ = ctx.get_state()
        ok = np.array_equal(r["X"], X1) and np.array_equal(r["lnp"], lnp1) and np.array_equal(r["nacc"], nacc1)
        print(("ok  " if ok else "FAIL"), "case", i, {k: c[k] for k in ("world", "parts", "block", "W", "P", "K", "big")}, flush=True)
        bad += not ok
    assert bad == 0
    print("soak ok")
