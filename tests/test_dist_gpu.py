"""N > 1 trainer paths on the GPU box: two ranks (gloo; they share the one GPU) whose sample counts differ -- rank 0
relabels hindsight records, rank 1 has none -- must issue the same number of gradient all-reduces, finish, and end with
identical replicas.  Same for the self-orientation agent when one rank has no orientation sample at all."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from twoarmy_amd import dist as twdist
    from twoarmy_amd.engine import TwoarmyEngine
    twdist.init_from_env(backend="gloo")
    torch.manual_seed(1234)                                   # identical replicas
    N, T = 24, 60
    eng = TwoarmyEngine(4, N, 17, device="cuda:0", seed=9981, env_id0=rank * N)
    calls = [0]
    if mode == "ppo":
        from twoarmy_amd.soa.agent.PPO import PPO
        from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
        torch.manual_seed(1234 + rank)                         # DIFFERENT replicas: the broadcast must make them equal
        agent = PPO()
        agent.K_epochs = 2
        agent.to("cuda:0").use_nhwc()                          # the entry points' order: to(device), channels-last, broadcast
        twdist.broadcast_parameters([agent.actor, agent.critic])
        bucket = twdist.GradBucket([list(agent.actor.parameters()), list(agent.critic.parameters())])
        agent.grad_sync = bucket
        tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=4096)          # minibatch > T * N
        tr.collect()
        if rank == 0:
            tr.relabel()                                       # rank 0: rollout + hindsight records; rank 1: rollout only
        n_her = 0 if tr.her is None else int(tr.her["t"].numel())
        tr.minibatch = 1024                                    # rank 0 needs more optimiser steps than rank 1
        tr.update()
        calls[0] = bucket.n_reduces
        assert bucket.n_copied == 0                            # zero-copy path: nothing was flattened / scattered
        nets = [agent.actor, agent.critic]
    else:
        from twoarmy_amd.soa.agent.Self_orientation_agent import self_orinetation_agent
        from twoarmy_amd.soa.soa_vec import VecSoATrainer
        agent = self_orinetation_agent()
        agent.K_epochs, agent.K_epochs_pre_agent_position = 1, 2
        agent.to("cuda:0")
        agent.grad_sync = twdist.GradBucket([list(agent.actor.parameters()), list(agent.critic.parameters())])
        ob = twdist.GradBucket(list(agent.agent_position_preditor.parameters()))
        agent.grad_sync_orient = ob
        tr = VecSoATrainer(agent, eng, rollout_steps=T, minibatch=1024, orient_minibatch=256)
        tr.collect()
        if rank == 0:
            tr.relabel()                                       # hindsight records feed the orientation head on rank 0 only
        n_her = 0 if tr.her is None else int(tr.her["t"].numel())
        if rank == 1:
            tr.term.zero_()                                    # rank 1: no success and no hindsight record -> zero samples
        tr.update()
        calls[0] = ob.n_reduces
        assert ob.n_copied == 0 and agent.grad_sync.n_copied == 0
        nets = [agent.agent_position_preditor]
    torch.cuda.synchronize()
    chk = [float(p.detach().double().sum()) for net in nets for p in net.parameters()]
    q.put((rank, calls[0], n_her, chk))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("mode", ["ppo", "soa"])
def test_ranks_with_different_sample_counts_stay_in_step(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, c0, h0, k0), (r1, c1, h1, k1) = res
    assert h0 > 0 and h1 == 0                                  # the ranks really had different sample counts
    assert c0 == c1 and c0 > 0                                 # ... and still issued the same number of all-reduces
    assert k0 == k1                                            # replicas identical after the update


def _copy_kernel_worker(rank, world, port, q):
    """Kernel names of one optimiser step without a bucket and of one with the two-group bucket on two ranks: the
    bucket must not add a single copy kernel (gradients are views into the flat buffer; what it adds is one fill, the
    in-place accumulations autograd does anyway, and one scale)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from torch.profiler import ProfilerActivity, profile
    from twoarmy_amd import dist as twdist
    from twoarmy_amd.soa.agent.PPO import PPO
    twdist.init_from_env(backend="gloo")
    torch.manual_seed(1234)
    agent = PPO()
    agent.to("cuda:0").use_nhwc()
    B = 256
    g = torch.Generator(device="cuda").manual_seed(5 + rank)
    x = torch.randn(B, 4, 289, device="cuda", generator=g)
    p4 = torch.randn(B, 4, 2, device="cuda", generator=g)
    goal = torch.randn(B, 2, device="cuda", generator=g)
    a = torch.randint(0, 5, (B,), device="cuda", dtype=torch.int32)
    lp = torch.full((B, 1), -1.6, device="cuda")
    adv, tv = torch.randn(B, 1, device="cuda", generator=g), torch.randn(B, 1, device="cuda", generator=g)

    def kernels():
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            agent.minibatch_step_x(x, p4, goal, a, lp, adv, tv)
            torch.cuda.synchronize()
        return [e.name for e in prof.events() if str(e.device_type).endswith("CUDA") and "memcpy" not in e.name.lower()]
    agent.minibatch_step_x(x, p4, goal, a, lp, adv, tv)        # warm-up (MIOpen search, Adam state)
    plain = kernels()
    bucket = twdist.GradBucket([list(agent.actor.parameters()), list(agent.critic.parameters())])
    agent.grad_sync = bucket
    agent.minibatch_step_x(x, p4, goal, a, lp, adv, tv)
    with_bucket = kernels()
    is_copy = lambda n: "direct_copy" in n or "copy_kernel" in n.lower() or "CatArrayBatchedCopy" in n    # noqa: E731
    q.put((rank, sum(map(is_copy, plain)), sum(map(is_copy, with_bucket)), len(plain), len(with_bucket), bucket.n_reduces,
           bucket.n_copied))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_adds_no_copy_kernels():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_copy_kernel_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, copies_plain, copies_bucket, n_plain, n_bucket, n_reduces, n_copied in res:
        assert copies_bucket <= copies_plain, (copies_plain, copies_bucket)     # not one copy kernel more than without a bucket
        assert n_reduces == 4 and n_copied == 0                                  # 2 steps x (actor, critic)
        assert n_plain > 20 and n_bucket > 20                                    # the profiler really saw the step's kernels
