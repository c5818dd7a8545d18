"""CPU restatement of the reference's hindsight relabelling (Buffer_gridworld.her_func,
soa/env_buffer.py:101-143) over a time-major rollout, producing index records instead of buffer copies.

TEST INFRASTRUCTURE ONLY (checker for ppo_her_relabel in <package>/csrc/ppo_kernels.hip).
Pinned against tests/golden/her.npz (buffers recorded from the reference's own her_func): materialising
the index records of this file reproduces the records the reference appended (tests/test_her_cpu.py).
The default pick stream (Philox Fisher-Yates, include/twoarmy_ppo.h) replaces the reference's global
np.random.choice; explicit `choices` replay any stream, which is how the golden pin works.
"""
import numpy as np

from philox import philox4x32_10

HER_TAG = 0x54574F48
MAX_LEN = 64


def first_visit(yx):
    """np.unique(p[:, 4, 0:2], return_index=True, axis=0) of env_buffer.py:107 -- literally."""
    _, idx = np.unique(np.asarray(yx), return_index=True, axis=0)
    return idx


def philox_picks(seed, env_id, t1, U, k):
    """Positions in the sorted unique array: partial Fisher-Yates, pick j swaps perm[j] <-> perm[j + w_j % (U - j)]."""
    w = [int(x) for x in philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, env_id & 0xFFFFFFFF,
                                       t1 & 0xFFFFFFFF, 0, HER_TAG)]
    perm = list(range(max(U, 1)))
    for j in range(min(k, 4)):
        s = j + w[j] % (U - j)
        perm[j], perm[s] = perm[s], perm[j]
    return perm[:k]


def relabel(pos, terminated, truncated, age0, reward, choices=None, seed=0, env_id0=0, step0=0, max_goals=4, skip=0):
    """pos [T,N,2] achieved (y,x) after each step; returns dict(t, n, goal, reward, done, counts) in the order
    env-major / episode time order / pick order / prefix time order.

    skip = 0: her_func on 5-frame records (env_buffer.py:101-143).  skip = 4: pre_her_func / pre_f_her_func on the
    9-frame window records (env_buffer.py:145-280): record i of an episode is stored four steps late and its newest
    frame is the state after step i + 4, so the first visits are taken among the states after steps 4, 5, ... only,
    `0 < index` excludes the first of THOSE, and a pick relabels the transitions 0 .. index + 4 (prefix records
    0 .. index plus the four sliding tail windows).  The four repeats of the terminal state at the end of the
    reference's record sequence are never first visits (episodes here are longer than four steps)."""
    T, N = terminated.shape
    done = (np.asarray(terminated) | np.asarray(truncated)) != 0
    ot, on, og, orw, od = [], [], [], [], []
    counts = np.zeros(N, np.int32)
    for n in range(N):
        start = 0 if int(age0[n]) == 0 else -1
        for t1 in range(T):
            if not done[t1, n]:
                continue
            s0, start = start, t1 + 1
            if s0 < 0 or t1 - s0 + 1 > MAX_LEN:
                continue
            ep = pos[s0:t1 + 1, n]                                   # the episode's records (env_buffer.py:105-106)
            if ep.shape[0] <= skip:
                continue
            fv = first_visit(ep[skip:])
            U = fv.size
            k = min(max_goals, U)                                    # env_buffer.py:109-112
            if choices is None:
                picks = philox_picks(seed, env_id0 + n, step0 + t1, U, k)
            else:
                picks = [int(c) for c in choices[t1, n, :k]]
            for pk in picks:
                if not (0 <= pk < U):
                    continue
                index = int(fv[pk])
                if not index > 0:                                    # `if 0 < index < cap` (env_buffer.py:119, :163)
                    continue
                index += skip                                        # window record `index` <-> transition index + 4
                for i in range(index + 1):
                    ot.append(s0 + i); on.append(n); og.append(ep[index])
                    orw.append(np.float32(0.9) if i == index else reward[s0 + i, n])
                    od.append(1 if i == index else 0)
                counts[n] += index + 1
    return dict(t=np.array(ot, np.int32), n=np.array(on, np.int32),
                goal=np.array(og, np.float32).reshape(-1, 2), reward=np.array(orw, np.float32),
                done=np.array(od, np.uint8), counts=counts)
