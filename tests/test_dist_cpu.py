"""N>1 path on CPU (gloo): the zero-copy gradient bucket, env sharding, global advantage statistics -- world sizes 2
and 8 -- and a VecPPOTrainer-shaped update on 8 ranks whose sample counts differ (skewed hindsight-record counts)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(target, world, extra=(), timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=timeout) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.set_num_threads(1)
    from twoarmy_amd import dist as twdist
    r, w, _ = twdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    return twdist


def _worker(rank, world, port, q):
    twdist = _init(rank, world, port)
    torch.manual_seed(1234 + rank)                      # ranks start with DIFFERENT weights on purpose
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    twdist.broadcast_parameters([net])
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    bucket = twdist.GradBucket(list(net.parameters()))
    x = torch.full((4, 7), float(rank + 1))
    # zero-copy protocol: zero() -> backward accumulates straight into the bucket -> reduce -> finish
    bucket.zero()
    net(x).sum().backward()
    assert all(p.grad is v for p, v in zip(bucket.params, bucket.views))          # autograd kept the views
    assert all(p.grad.untyped_storage().data_ptr() == bucket.flat.untyped_storage().data_ptr() for p in bucket.params)
    local = bucket.flat.clone()
    bucket()
    synced = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    ok = bool(torch.allclose(synced, expect, atol=1e-6)) and bucket.n_copied == 0 and bucket.n_reduces == 1
    # legacy protocol: the optimiser's own zero_grad() detaches the views; bucket() copies the fresh gradients in
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.zero_grad()
    net(x).sum().backward()
    bucket()
    synced2 = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    ok = ok and bool(torch.allclose(synced2, expect, atol=1e-6)) and bucket.n_copied == len(bucket.params)
    lo, hi = twdist.shard_range(4099, rank, world)
    q.put((rank, ok, w0.tolist(), (lo, hi), bucket.numel))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_bucket_allreduce(world):
    res = _spawn(_worker, world)
    assert all(r[1] for r in res)
    assert all(r[2] == res[0][2] for r in res)          # broadcast made the replicas identical
    if world == 2:
        assert res[0][3] == (0, 2050) and res[1][3] == (2050, 4099)
    assert res[0][3][0] == 0 and res[-1][3][1] == 4099 and all(res[i][3][1] == res[i + 1][3][0] for i in range(world - 1))
    assert res[0][4] == 7 * 5 + 5 + 5 * 3 + 3


def test_shard_range_partitions_exactly():
    from twoarmy_amd.dist import shard_range
    for total in (4096, 8192, 16384, 4099, 7):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_bucket_views_follow_channels_last_parameters():
    """Conv weights switched to channels-last (use_nhwc) get channels-last gradient views of the flat buffer; backward
    accumulates into them in place (no process group needed)."""
    from twoarmy_amd.dist import GradBucket
    conv = torch.nn.Conv2d(4, 8, 3).to(memory_format=torch.channels_last)
    lin = torch.nn.Linear(8, 2)
    b = GradBucket([list(conv.parameters()), list(lin.parameters())])
    assert b.numel == 4 * 8 * 9 + 8 + 16 + 2 and len(b.parts) == 2 and b.parts[0].numel() == 4 * 8 * 9 + 8
    assert conv.weight.grad.stride() == conv.weight.stride()
    b.zero()
    y = conv(torch.randn(2, 4, 6, 6)).mean((2, 3))
    lin(y).sum().backward()
    assert conv.weight.grad is b.views[0] and float(b.parts[0].abs().sum()) > 0 and float(b.parts[1].abs().sum()) > 0
    b.zero()
    assert float(b.flat.abs().sum()) == 0 and float(conv.weight.grad.abs().sum()) == 0


def _norm_worker(rank, world, port, q):
    twdist = _init(rank, world, port)
    g = torch.Generator().manual_seed(7)
    full = torch.randn(1000, generator=g) * 3 + 1
    cuts = [0] + [100 + 113 * r for r in range(world - 1)] + [1000]               # unequal shards
    mine = full[cuts[rank]:cuts[rank + 1]].clone()
    twdist.global_adv_norm_(mine)
    q.put((rank, mine.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_global_advantage_normalisation(world):
    res = _spawn(_norm_worker, world)
    g = torch.Generator().manual_seed(7)
    full = torch.randn(1000, generator=g) * 3 + 1
    want = (full - full.mean()) / (full.std() + 1e-8)                   # torch unbiased std, like PPO.py:115
    got = torch.tensor(sum((r[1] for r in res), []))
    assert torch.allclose(got, want, atol=1e-5)


def _update_worker(rank, world, port, q):
    """The optimiser-step protocol of VecPPOTrainer.update / PPO.minibatch_step_x on CPU stand-ins: two networks with
    their own Adam, a two-group GradBucket (the "actor" group is reduced while the "critic" backward runs), K epochs over
    a sample set whose size differs per rank (rollout samples + a skewed number of hindsight records)."""
    twdist = _init(rank, world, port)
    from twoarmy_amd.soa.ppo_vec import agree_on_steps, padded_permutation
    torch.manual_seed(99)
    actor = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 5))
    critic = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    twdist.broadcast_parameters([actor, critic])
    oa, oc = torch.optim.Adam(actor.parameters(), 1e-3), torch.optim.Adam(critic.parameters(), 1e-3)
    bucket = twdist.GradBucket([list(actor.parameters()), list(critic.parameters())])
    minibatch, K = 64, 3
    total = 512 + (0, 700, 13, 0, 257, 64, 1, 333)[rank % 8]              # rollout + skewed hindsight-record counts
    g = torch.Generator().manual_seed(1000 + rank)
    x, y = torch.randn(total, 6, generator=g), torch.randn(total, 1, generator=g)
    local_steps = -(-total // minibatch)
    n_steps = agree_on_steps(local_steps, "cpu")
    done = 0
    for ep in range(K):
        perm = padded_permutation(torch.randperm(total, generator=g), n_steps, minibatch)
        assert -(-perm.numel() // minibatch) == n_steps
        for i in range(0, perm.numel(), minibatch):
            idx = perm[i:i + minibatch]
            la = torch.log_softmax(actor(x[idx]), 1).mean()
            lv = torch.nn.functional.smooth_l1_loss(critic(x[idx]), y[idx])
            bucket.zero()
            la.backward()
            bucket.reduce_async(0)                      # in flight while the critic's backward runs
            lv.backward()
            bucket.reduce_async(1)
            bucket.finish()
            oa.step(); oc.step()
            done += 1
    chk = [float(p.detach().double().sum()) for net in (actor, critic) for p in net.parameters()]
    q.put((rank, done, n_steps, local_steps, bucket.n_reduces, bucket.n_copied, chk))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_shaped_update_world8_skewed_sample_counts():
    res = _spawn(_update_worker, 8)
    steps = {r[2] for r in res}
    assert len(steps) == 1 and steps.pop() == max(r[3] for r in res)      # everyone takes the largest rank's step count
    assert len({r[3] for r in res}) > 3                                    # ... although their own counts differ
    assert all(r[1] == 3 * r[2] for r in res)
    assert all(r[4] == 2 * r[1] and r[5] == 0 for r in res)                # two all-reduces per step, nothing copied
    assert all(r[6] == res[0][6] for r in res)                             # replicas identical after the update
