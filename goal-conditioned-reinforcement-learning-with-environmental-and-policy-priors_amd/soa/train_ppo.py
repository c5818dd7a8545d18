"""Entry point of the PPO baseline -- same flag names as the reference's soa/train_ppo.py:23-40, driving the
vectorised HIP engine instead of one Python env.

  python -m twoarmy_amd.soa.train_ppo --env MiniGrid-twoarmy-17x17-v6 --num_envs 4096 --updates 10
  python -m torch.distributed.run --nproc-per-node 8 ... train_ppo.py --num_envs 8192      (envs sharded per rank)

Reference-only flags that concerned rendering / absolute log paths are accepted and ignored.
"""
import argparse
import os
import random
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--env", default="MiniGrid-twoarmy-17x17-v4")
    p.add_argument("--seed", type=int, default=9981)
    p.add_argument("--tile_size", type=int, default=17)
    p.add_argument("--batch_size", type=int, default=128, help="reference minibatch for its 2048-record buffer; "
                   "the vectorised loop uses --minibatch")
    p.add_argument("--her", default=True)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--lr", type=float, default=0.0001)
    p.add_argument("--weight_decay", type=float, default=0.0001)
    p.add_argument("--lr_gamma", type=float, default=0.8)
    p.add_argument("--lr_step_size", type=int, default=200)
    p.add_argument("--track_buffer_file", default=None)
    p.add_argument("--num_episodes", type=int, default=1000000)
    p.add_argument("--max_steps", type=int, default=50)
    p.add_argument("--log_dir", default=None)
    p.add_argument("--cuda", default="cuda:0")
    p.add_argument("--server", default=True)
    # vectorised-engine options (no reference counterpart)
    p.add_argument("--num_envs", type=int, default=4096, help="total envs over all ranks")
    p.add_argument("--rollout_steps", type=int, default=128)
    p.add_argument("--minibatch", type=int, default=4096)
    p.add_argument("--updates", type=int, default=1)
    p.add_argument("--amp", default="fp32", choices=["fp32", "bf16"],
                   help="GEMM dtype of the actor-critic (bf16 = torch.autocast, no parity claim; fp32 = reference)")
    p.add_argument("--conv_layout", default="nhwc", choices=["nhwc", "nchw"],
                   help="nhwc: channels-last conv stacks + fused bias/ReLU epilogues (fast on MI355X); nchw: the literal "
                        "nn.Sequential path")
    p.add_argument("--graph_rollout", default="auto", choices=["auto", "on", "off"],
                   help="replay the rollout (actor in the loop) as one HIP graph; auto = on for <= 512 envs per rank "
                        "(plain PPO agent), where the ~25 launches per step are host-bound (1.7x at 256 envs)")
    p.add_argument("--frame_codes", action="store_true", help="store rollout frames as uint8 codes (4x smaller, exact)")
    p.add_argument("--miopen_benchmark", action="store_true",
                   help="torch.backends.cudnn.benchmark = True: MIOpen benchmarks its solvers once per conv shape (minutes at "
                        "start-up, cached on disk) instead of taking its heuristic pick: epoch 1.66 -> 1.56 s at 4096 envs")
    p.add_argument("--k_epochs", type=int, default=10)
    p.add_argument("--k_epochs_orientation", type=int, default=50, help="SoA: epochs of the orientation head per update")
    p.add_argument("--gae_lambda", type=float, default=0.0)
    p.add_argument("--normalize_adv", action="store_true")
    p.add_argument("--predictor_file", default=None, help="checkpoint with model_encoder / model_decoder / "
                   "model_predictor (train_ppo_predictor.py:38,81-85); random-init world model when absent")
    return p


def main(argv=None, predictor=False, soa=False):
    args = build_parser().parse_args(argv)
    from .. import dist as twdist
    from ..engine import TwoarmyEngine
    from .agent.PPO import PPO
    from .ppo_vec import VecPPOTrainer

    rank, world, local_rank = twdist.init_from_env()
    seed = None if args.seed == -1 else args.seed
    random.seed(seed); np.random.seed(seed); os.environ["PYTHONHASHSEED"] = str(seed)
    if seed is not None:
        torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count())) if world > 1 else torch.device(args.cuda)
    torch.cuda.set_device(device)
    if args.miopen_benchmark:
        torch.backends.cudnn.benchmark = True

    if soa:
        from .agent.Self_orientation_agent import self_orinetation_agent
        from .soa_vec import VecSoATrainer
        agent = self_orinetation_agent(log_root=args.log_dir)
        agent.K_epochs_pre_agent_position = args.k_epochs_orientation
        if args.predictor_file:
            agent.load_world_model(torch.load(args.predictor_file, map_location="cpu", weights_only=True))
    elif predictor:
        from .agent.PPO_Predictor import ppo_predictor
        agent = ppo_predictor(log_root=args.log_dir)
        if args.predictor_file:
            agent.load_world_model(torch.load(args.predictor_file, map_location="cpu", weights_only=True))
    else:
        agent = PPO(log_root=args.log_dir)
    agent.name = "%s_%s_%sseed_" % ("soa" if soa else ("ppo_predictor" if predictor else "ppo"), args.env, seed)
    agent.gamma, agent.K_epochs = args.gamma, args.k_epochs
    agent.gae_lambda, agent.use_done_mask, agent.normalize_adv = args.gae_lambda, args.gae_lambda > 0, args.normalize_adv
    agent.sample_seed = (seed or 0) + 7919 * rank
    agent.amp_dtype = torch.bfloat16 if args.amp == "bf16" else None
    agent.to(device)
    if args.conv_layout == "nhwc":
        agent.use_nhwc()
    twdist.broadcast_parameters([agent.actor, agent.critic] +
                                ([agent.encoder, agent.decoder, agent.predictor] if (predictor or soa) else []) +
                                ([agent.agent_position_preditor] if soa else []))
    if world > 1:
        agent.grad_sync = twdist.GradBucket([list(agent.actor.parameters()), list(agent.critic.parameters())])
        if soa:                                  # the orientation head has its own optimiser step, hence its own bucket
            agent.grad_sync_orient = twdist.GradBucket(list(agent.agent_position_preditor.parameters()))

    lo, hi = twdist.shard_range(args.num_envs, rank, world)
    variant = 4 if args.env.endswith("v4") else 6
    engine = TwoarmyEngine(variant, hi - lo, 17, device=device, seed=seed or 0, env_id0=lo, max_steps=args.max_steps)
    Trainer = VecSoATrainer if soa else VecPPOTrainer
    trainer = Trainer(agent, engine, args.rollout_steps, args.minibatch, frame_codes=args.frame_codes)
    trainer.use_graph = (not predictor and not soa) and (args.graph_rollout == "on" or
                                                         (args.graph_rollout == "auto" and hi - lo <= 512))
    her = str(args.her).lower() not in ("false", "0", "no")
    score = 0.0
    for u in range(args.updates):
        t0 = time.perf_counter()
        trainer.collect()
        her = trainer.her_switch(her, score)                  # train_ppo.py:128-131 of the reference
        if her and agent.gae_lambda == 0.0:
            trainer.relabel()
        score = trainer.running_score(score)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_her = 0 if trainer.her is None else int(trainer.her["t"].numel())
        la, lv = trainer.update()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = trainer.stats()
        trainer.carry_over()
        if rank == 0:
            print("update %d: rollout %.3fs (%.0f env-steps/s/rank) update %.3fs action_loss %.5f value_loss %.5f "
                  "episodes %d successes %d mean_r %.4f her_records %d score %.4f" % (u, t1 - t0, trainer.T * trainer.N / (t1 - t0), t2 - t1,
                                                            float(la), float(lv), st["episodes"], st["successes"],
                                                            st["mean_reward"], n_her, score), flush=True)
    engine.close()
    return trainer


if __name__ == "__main__":
    main()
