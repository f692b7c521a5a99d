"""N>1 path on CPU: world_size-2 gloo processes exercise shard ranges, the end-of-rollout gather
(all_gather_into_tensor / uneven all_gather) and the info reduction, and check that the keyed RNG of
the oracle is independent of how envs are sharded."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mapdn_amd.sharding import shard_range


def test_shard_ranges_cover_exactly():
    for total in (1, 7, 4096, 8192, 65536, 1000):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, world, r) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapdn_amd.sharding import gather_rollout, reduce_mean_info, shard_range
    from oracle import philox
    lo, hi = shard_range(total, world, rank)
    # per-env "episode return": a pure function of the GLOBAL env id (as the Philox keying guarantees)
    ret = torch.tensor([philox.normals(0, e, 0, 0, 2)[0] for e in range(lo, hi)], dtype=torch.float64)
    sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
    full = gather_rollout(ret, sizes)
    info = torch.arange(lo, hi, dtype=torch.float64)[:, None].repeat(1, 11)
    mean = reduce_mean_info(info, total)
    even = gather_rollout(torch.full((4, 3), float(rank)))
    q.put((rank, full.numpy(), mean.numpy(), even.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 9])
def test_gloo_world2_gather(total):
    from oracle import philox
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([philox.normals(0, e, 0, 0, 2)[0] for e in range(total)])
    for rank, full, mean, even in got:
        assert np.array_equal(full, want)                       # global env-id order, identical on every rank
        assert np.allclose(mean, np.arange(total).mean())
        assert even.shape == (8, 3) and (even[:4] == 0).all() and (even[4:] == 1).all()
