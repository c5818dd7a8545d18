"""Data collection for the offline world-model training (reference soa/datacol_predictor.py:60-171): a uniformly
random policy (`np.random.choice(range(5))` there, the engine's Philox action stream here) is rolled out and every
step becomes one 9-frame window record -- 3 history frames, the acting state at index 3, 5 future frames, the
terminal state repeated after the episode end, the reset frame before its start -- with the reference's structured
dtype.  The reference steps one Python env; here the N envs of the HIP engine produce the frames time-major in one
launch per rollout and the windows are index arithmetic on them.

  python -m twoarmy_amd.soa.datacol_predictor --env MiniGrid-twoarmy-17x17-v6 --buffer_pre_capacity 10000 --log_dir out/
"""
import argparse
import os
from datetime import datetime

import numpy as np
import torch

GOAL_YX = (2.0, 14.0)
INIT_YX = (15.0, 3.0)


def pre_transition_dtype(grid_cells=289):
    """The record layout of datacol_predictor.py:84-86."""
    return np.dtype([("s", np.float64, (9, grid_cells)), ("a", np.int64, (5, 1)), ("p", np.float64, (9, 2)),
                     ("g", np.float64, (2,)), ("r", np.float64, (5, 1)), ("d", np.int64, (5, 1)),
                     ("a_logp", np.float64, (5, 1))])


@torch.no_grad()
def windows_from_rollout(frames, pos, action, reward, term, trunc, init_frame):
    """Time-major rollout that starts with all envs freshly reset -> dict of window tensors, one row per usable step.

    frames [T,N,289] / pos [T,N,2]: state after every step; init_frame [289]: the reset state.  State index s means
    "before step s"; state 0 is the reset state, state s > 0 is frames[s-1] unless an episode started at s."""
    T, N = term.shape
    dev = frames.device
    done = (term | trunc) != 0
    tt = torch.arange(T, device=dev).view(T, 1).expand(T, N)
    big = torch.full((T, N), 1 << 20, device=dev, dtype=torch.long)
    end = torch.flip(torch.cummin(torch.flip(torch.where(done, tt, big), [0]), 0).values, [0])       # episode end step
    # episode start (state index) of every step: the step after the previous done, 0 at the beginning
    prev_done = torch.cummax(torch.where(done, tt + 1, torch.zeros_like(tt)), 0).values
    start = torch.cat([torch.zeros((1, N), dtype=torch.long, device=dev), prev_done[:-1]], 0)
    ended = end < T
    usable = ended | (tt + 5 <= T)                      # 5 future states must exist (or the episode ends before)
    t_idx, n_idx = torch.nonzero(usable, as_tuple=True)
    u, s0 = end[t_idx, n_idx], start[t_idx, n_idx]
    k9 = torch.arange(9, device=dev).view(1, 9)
    w = t_idx.view(-1, 1) - 3 + k9                                           # state indices of the window
    w = torch.minimum(w, (u + 1).view(-1, 1))                                 # terminal state repeats (:151-163)
    is_init = w <= s0.view(-1, 1)                                             # before / at the episode start: reset state
    f_idx = (w - 1).clamp(0, T - 1)
    nn = n_idx.view(-1, 1).expand(-1, 9)
    s = torch.where(is_init.unsqueeze(-1), init_frame.view(1, 1, -1), frames[f_idx, nn])
    p = torch.where(is_init.unsqueeze(-1), torch.tensor(INIT_YX, device=dev).view(1, 1, 2), pos[f_idx, nn])
    k5 = torch.arange(5, device=dev).view(1, 5)
    q = torch.minimum(t_idx.view(-1, 1) + k5, u.view(-1, 1)).clamp(max=T - 1)  # steps of the 5-entry columns
    n5 = n_idx.view(-1, 1).expand(-1, 5)
    return dict(t=t_idx, n=n_idx, s=s, p=p, a=action[q, n5], r=reward[q, n5], d=term[q, n5].long())


def collect_windows(engine, n_records, rollout_steps=128):
    """Random-policy window records (numpy structured array of `pre_transition_dtype`) from the vector engine."""
    buf = np.zeros(n_records, dtype=pre_transition_dtype())
    filled = 0
    T = int(rollout_steps)
    init_frame = None
    out = engine.alloc_outputs(T)                       # ONE slab for every rollout (native record layout -> the pipelined
    #                                                     kernel); a slab per iteration would map / unmap hundreds of 2 MiB
    #                                                     chunks each time and leave its address range reserved
    while filled < n_records:
        engine.reset()
        if init_frame is None:
            ty, _, _ = engine.get_state()
            m = torch.where(torch.tensor(ty[0] == 2), -0.9, torch.where(torch.tensor(ty[0] == 6), -0.5, 0.9)).float()
            m[15 * 17 + 3] = 0.3
            init_frame = m.to(engine.device)
        acts = engine.fill_actions(T)                   # the engine's Philox stream: uniform over the 5 policy actions
        engine.rollout(T, out, actions=acts, autoreset=True, policy_idx=True)
        w = windows_from_rollout(out["matrix"], out["pos"], acts, out["reward"], out["terminated"], out["truncated"],
                                 init_frame)
        k = min(n_records - filled, int(w["t"].numel()))
        sl = slice(filled, filled + k)
        # the engine emits float32; the reference's records hold the doubles 0.9 / -0.9 / -0.5 / 0.3 and -0.01 ... 0.9
        buf["s"][sl] = np.round(w["s"][:k].double().cpu().numpy(), 6)
        buf["p"][sl] = w["p"][:k].double().cpu().numpy()
        buf["a"][sl] = w["a"][:k].cpu().numpy()[..., None]
        buf["r"][sl] = np.round(w["r"][:k].double().cpu().numpy(), 6)[..., None]
        buf["d"][sl] = w["d"][:k].cpu().numpy()[..., None]
        buf["g"][sl] = GOAL_YX
        filled += k
    return buf


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--env", default="MiniGrid-twoarmy-17x17-v6")
    p.add_argument("--agent", default="random_")
    p.add_argument("--seed", type=int, default=2345)
    p.add_argument("--buffer_pre_capacity", type=int, default=10000)
    p.add_argument("--num_envs", type=int, default=256)
    p.add_argument("--rollout_steps", type=int, default=128)
    p.add_argument("--log_dir", default="./runs/predictor_data/")
    p.add_argument("--cuda", default="cuda:0")
    args = p.parse_args(argv)
    from ..engine import TwoarmyEngine
    dev = torch.device(args.cuda)
    engine = TwoarmyEngine(4 if args.env.endswith("v4") else 6, args.num_envs, 17, device=dev, seed=args.seed)
    buf = collect_windows(engine, args.buffer_pre_capacity, args.rollout_steps)
    engine.close()
    os.makedirs(args.log_dir, exist_ok=True)
    name = "%s_%s%d_prebuffer_" % (args.env, args.agent, args.buffer_pre_capacity)
    path = os.path.join(args.log_dir, "predictor_" + name + datetime.now().strftime("%Y_%m_%d_%H_%M_%S") + ".npy")
    np.save(path, buf)
    print("stored %d window records -> %s" % (len(buf), path))
    return path


if __name__ == "__main__":
    main()
