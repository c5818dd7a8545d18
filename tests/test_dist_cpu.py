"""N>1 path on CPU: world_size-2 gloo run of the gradient bucket all-reduce + env sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from twoarmy_amd import dist as twdist
    r, w, _ = twdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(1234 + rank)                      # ranks start with DIFFERENT weights on purpose
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    twdist.broadcast_parameters([net])
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    bucket = twdist.GradBucket(list(net.parameters()))
    x = torch.full((4, 7), float(rank + 1))
    net(x).sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    bucket()
    synced = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    lo, hi = twdist.shard_range(4099, rank, world)
    q.put((rank, bool(torch.allclose(synced, expect, atol=1e-6)), w0.tolist(), (lo, hi), bucket.numel))
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2]                       # broadcast made the replicas identical
    assert res[0][3] == (0, 2050) and res[1][3] == (2050, 4099)
    assert res[0][4] == 7 * 5 + 5 + 5 * 3 + 3


def test_shard_range_partitions_exactly():
    from twoarmy_amd.dist import shard_range
    for total in (4096, 8192, 16384, 4099, 7):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _norm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from twoarmy_amd import dist as twdist
    twdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(7)
    full = torch.randn(1000, generator=g) * 3 + 1
    mine = full[:300].clone() if rank == 0 else full[300:].clone()      # unequal shards
    twdist.global_adv_norm_(mine)
    q.put((rank, mine.tolist()))
    dist.destroy_process_group()


def test_global_advantage_normalisation_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_norm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(7)
    full = torch.randn(1000, generator=g) * 3 + 1
    want = (full - full.mean()) / (full.std() + 1e-8)                   # torch unbiased std, like PPO.py:115
    got = torch.tensor(res[0] + res[1])
    assert torch.allclose(got, want, atol=1e-5)
