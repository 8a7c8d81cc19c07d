"""N > 1 path on CPU: world_size-2 gloo run of the frame sharding + result gather used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ethzasl_brisk_amd import sharding


def test_shard_frames_partition():
    for n in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            for mode in ("block", "cyclic"):
                parts = [sharding.shard_frames(n, r, world, mode) for r in range(world)]
                assert sorted(sum(parts, [])) == list(range(n))
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _fake_results(rank, batch, cap, pitch):
    rng = np.random.default_rng(100 + rank)
    counts = rng.integers(0, cap + 1, batch).astype(np.int32)
    counts[0] = 0 if rank == 0 else cap           # empty and full frames
    kps = rng.normal(size=(batch, cap, 7)).astype(np.float32)
    desc = rng.integers(0, 256, (batch, cap, pitch), dtype=np.uint8)
    return counts, kps, desc


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, cap, pitch, strings = 5, 9, 64, 48
        c, k, d = _fake_results(rank, batch, cap, pitch)
        res = sharding.gather_results(torch.from_numpy(c), torch.from_numpy(k), torch.from_numpy(d), strings)
        if rank == 0:
            ok = len(res) == world
            for r in range(world):
                cr, kr, dr = _fake_results(r, batch, cap, pitch)
                exp_k = np.concatenate([kr[f, :cr[f]] for f in range(batch)])
                exp_d = np.concatenate([dr[f, :cr[f], :strings] for f in range(batch)])
                ok &= np.array_equal(res[r][0].numpy(), cr)
                ok &= np.array_equal(res[r][1].numpy(), exp_k) and np.array_equal(res[r][2].numpy(), exp_d)
            q.put(bool(ok))
        else:
            assert res is None
        # fixed-size variant used inside bench.py's timed region
        res2 = sharding.gather_results_padded(torch.from_numpy(c), torch.from_numpy(k), torch.from_numpy(d), strings, cap)
        if rank == 0:
            ac, gk, gd = res2
            ok = ac.shape == (world, batch)
            for r in range(world):
                cr, kr, dr = _fake_results(r, batch, cap, pitch)
                ok &= np.array_equal(ac[r].numpy(), cr) and np.array_equal(gk[r].numpy(), kr[:, :cap]) and np.array_equal(gd[r].numpy(), dr[:, :cap, :strings])
            q.put(bool(ok))
        else:
            assert res2 is None
        # asynchronous double-buffered pipeline (what bench.py runs): three batches back to back, buffers rewritten
        # between them the way the engine's result buffers are
        tc, tk, td = torch.from_numpy(c.copy()), torch.from_numpy(k.copy()), torch.from_numpy(d.copy())
        pg = sharding.PaddedGather(tc, tk, td, strings, cap)
        for it in range(3):
            tk.copy_(torch.from_numpy(k) + it)
            pg.start()
        res3 = pg.finish()
        if rank == 0:
            ac, gk, gd = res3
            ok = tuple(ac.shape) == (world, batch)
            for r in range(world):
                cr, kr, dr = _fake_results(r, batch, cap, pitch)
                ok &= np.array_equal(ac[r].numpy(), cr) and np.array_equal(gk[r].numpy(), kr[:, :cap] + 2) and np.array_equal(gd[r].numpy(), dr[:, :cap, :strings])
            q.put(bool(ok))
        else:
            assert res3 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_results_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True
    assert q.get() is True
    assert q.get() is True
