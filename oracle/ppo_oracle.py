"""CPU restatement of the PPO math of the reference (soa/agent/PPO.py) in numpy float32.

TEST INFRASTRUCTURE ONLY (checker for <package>/csrc/ppo_kernels.hip).  Pinned against
tests/golden/ppo.npz (log-probs, entropies and update losses recorded from the reference's own
torch code).  GAE with lambda > 0 / done masks and advantage normalisation have no reference
counterpart: for those this file is the only pin ("parity unpinned" beyond the
lambda = 0 collapse, which equals PPO.py:113-114).
"""
import numpy as np

EPS = np.float32(np.finfo(np.float32).eps)
F = np.float32


def categorical(probs):
    """torch.distributions.Categorical(probs=p): normalised q, clamped logits, entropy (PPO.py:124-126)."""
    p = np.asarray(probs, F)
    q = p / p.sum(-1, keepdims=True, dtype=F)
    logits = np.log(np.clip(q, EPS, F(1) - EPS)).astype(F)
    ent = -(q * logits).sum(-1, dtype=F)
    return q, logits, ent


def sample(probs, uniforms):
    """Inverse-CDF sample as specified in include/twoarmy_ppo.h; returns (action, logp)."""
    q, logits, _ = categorical(probs)
    B, A = q.shape
    a = np.full(B, A - 1, np.int32)
    cum = np.zeros(B, F)
    found = np.zeros(B, bool)
    for k in range(A):
        cum = (cum + q[:, k]).astype(F)
        hit = (~found) & (cum > np.asarray(uniforms, F))
        a[hit] = k
        found |= hit
    return a, logits[np.arange(B), a]


def gae(reward, value, next_value, done, gamma, lam, use_done_mask):
    """Sequential float32 recurrence (the kernel uses a parallel scan: compare with a tolerance for lam > 0)."""
    r, v, nv = (np.asarray(x, F) for x in (reward, value, next_value))
    T, N = r.shape
    cut = (1 - np.asarray(done, F)) if use_done_mask else np.ones((T, N), F)
    g = F(gamma)
    target = (r + (g * nv).astype(F) * cut).astype(F)          # PPO.py:113 (same rounding order)
    delta = (target - v).astype(F)                             # PPO.py:114
    adv = np.zeros((T, N), F)
    nxt = np.zeros(N, F)
    coef = F(gamma) * F(lam)
    for t in range(T - 1, -1, -1):
        nxt = (delta[t] + coef * cut[t] * nxt).astype(F)
        adv[t] = nxt
    return adv, target, (adv + v).astype(F)


def adv_norm(adv, eps=1e-8):
    a = np.asarray(adv, np.float64)
    return ((a - a.mean()) / (a.std(ddof=1) + eps)).astype(F)   # torch .std() is unbiased


def losses(probs, action, old_logp, adv, value, target_v, clip=0.1, ent_coef=0.01):
    """(action_loss, value_loss) of PPO.py:124-133, float32 like the reference."""
    q, logits, ent = categorical(probs)
    B = q.shape[0]
    logp = logits[np.arange(B), np.asarray(action)]
    ratio = np.exp((logp - np.asarray(old_logp, F)).astype(F)).astype(F)
    ad = np.asarray(adv, F)
    s1 = ratio * ad
    s2 = np.clip(ratio, F(1 - clip), F(1 + clip)) * ad
    al = (-np.minimum(s1, s2) - F(ent_coef) * ent).astype(F).mean(dtype=F)
    d = np.asarray(value, F) - np.asarray(target_v, F)
    ab = np.abs(d)
    vl = np.where(ab < 1, F(0.5) * d * d, ab - F(0.5)).astype(F).mean(dtype=F)
    return F(al), F(vl)


def gather_stack(frames, pos_frames, k_idx, n_idx, age, init_frame, init_pos):
    B = len(k_idx)
    out = np.empty((B, 4, frames.shape[-1]), F)
    pos = np.empty((B, 4, 2), F)
    for b in range(B):
        for j in range(4):
            back = 3 - j
            if age[b] - back <= 0:
                out[b, j], pos[b, j] = init_frame, init_pos
            else:
                out[b, j] = frames[k_idx[b] - back, n_idx[b]]
                pos[b, j] = pos_frames[k_idx[b] - back, n_idx[b]]
    return out, pos
