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
        bucket = twdist.GradBucket(list(agent.actor.parameters()) + list(agent.critic.parameters()))

        def sync(_p=None):
            calls[0] += 1
            bucket()
        agent.grad_sync = sync
        tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=4096)          # minibatch > T * N
        tr.collect()
        if rank == 0:
            tr.relabel()                                       # rank 0: rollout + hindsight records; rank 1: rollout only
        n_her = 0 if tr.her is None else int(tr.her["t"].numel())
        tr.minibatch = 1024                                    # rank 0 needs more optimiser steps than rank 1
        tr.update()
        nets = [agent.actor, agent.critic]
    else:
        from twoarmy_amd.soa.agent.Self_orientation_agent import self_orinetation_agent
        from twoarmy_amd.soa.soa_vec import VecSoATrainer
        agent = self_orinetation_agent()
        agent.K_epochs, agent.K_epochs_pre_agent_position = 1, 2
        agent.to("cuda:0")
        agent.grad_sync = twdist.GradBucket(list(agent.actor.parameters()) + list(agent.critic.parameters()))
        ob = twdist.GradBucket(list(agent.agent_position_preditor.parameters()))

        def sync_o(_p=None):
            calls[0] += 1
            ob()
        agent.grad_sync_orient = sync_o
        tr = VecSoATrainer(agent, eng, rollout_steps=T, minibatch=1024, orient_minibatch=256)
        tr.collect()
        if rank == 0:
            tr.relabel()                                       # hindsight records feed the orientation head on rank 0 only
        n_her = 0 if tr.her is None else int(tr.her["t"].numel())
        if rank == 1:
            tr.term.zero_()                                    # rank 1: no success and no hindsight record -> zero samples
        tr.update()
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
