"""Device hindsight relabelling (ppo_her_relabel, C ABI) vs oracle/her_oracle.py -- which tests/test_her_cpu.py pins
to the buffers the reference's own her_func produced -- and directly against those reference buffers."""
import numpy as np
import pytest
import torch

import her_oracle
import twoarmy_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
SEED = 9981


def _ops():
    from twoarmy_amd import ppo_ops
    return ppo_ops


def _dev(rec):
    return {k: v.cpu().numpy() for k, v in rec.items()}


def _assert_same(got, want):
    for k in ("counts", "t", "n", "goal", "reward", "done"):
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, (k, got[k].shape, want[k].shape)
        assert np.array_equal(got[k], want[k]), k


def _rollout(variant, N, T, env0=0):
    ref = orc.rollout(variant, N, T, SEED, env0=env0, view=3, want_obs=False, want_matrix=False)
    return ref["pos"], ref["terminated"], ref["truncated"], ref["reward"]


@pytest.mark.parametrize("variant,N,T", [(6, 300, 128), (4, 257, 200), (6, 1, 60), (4, 64, 49)])
@pytest.mark.parametrize("mode", ["choices", "philox"])
def test_her_records_match_oracle(variant, N, T, mode):
    pos, term, trunc, rew = _rollout(variant, N, T)
    rs = np.random.RandomState(N + T)
    age0 = np.where(rs.rand(N) < 0.3, rs.randint(1, 20, N), 0).astype(np.int32)    # some envs start mid-episode
    choices = None
    if mode == "choices":                     # arbitrary external picks, incl. out-of-range and repeated entries
        choices = rs.randint(-1, 12, size=(T, N, 4)).astype(np.int32)
    want = her_oracle.relabel(pos, term, trunc, age0, rew, choices=choices, seed=SEED, env_id0=5, step0=1000)
    got = _ops().her_relabel(torch.tensor(pos, device=DEV), torch.tensor(term, device=DEV),
                             torch.tensor(trunc, device=DEV), torch.tensor(age0, device=DEV),
                             torch.tensor(rew, device=DEV),
                             None if choices is None else torch.tensor(choices, device=DEV),
                             seed=SEED, env_id0=5, step0=1000)
    _assert_same(_dev(got), want)
    if N >= 64 and T >= 100:
        assert want["t"].size > 0 and int(want["done"].sum()) > 0


def test_her_against_reference_buffers(golden_dir):
    """The records the reference's her_func appended (tests/golden/her.npz), reproduced by the device kernel with
    the reference's np.random picks replayed through `choices`."""
    z = np.load(golden_dir + "/her.npz")
    checked = 0
    for ci in range(int(z["n_cases"])):
        cap, seed, pre, L, cnt_before, _, cnt_after, _, _ = (int(v) for v in z["c%d_meta" % ci])
        if cnt_before <= pre:
            continue
        n_rec = cnt_before - pre
        pos = np.ascontiguousarray(z["c%d_before_p" % ci][pre:cnt_before][:, 4, 0:2].reshape(n_rec, 1, 2))
        rew = np.ascontiguousarray(z["c%d_before_r" % ci][pre:cnt_before].reshape(n_rec, 1))
        term = np.zeros((n_rec, 1), np.uint8); term[-1] = 1
        fv = her_oracle.first_visit(pos[:, 0])
        np.random.seed(seed)
        chosen = np.random.choice(fv, size=min(4, fv.size), replace=False)
        choices = np.full((n_rec, 1, 4), -1, np.int32)
        choices[-1, 0, :chosen.size] = [int(np.where(fv == c)[0][0]) for c in chosen]
        got = _dev(_ops().her_relabel(torch.tensor(pos, device=DEV), torch.tensor(term, device=DEV),
                                      torch.zeros((n_rec, 1), dtype=torch.uint8, device=DEV),
                                      torch.zeros(1, dtype=torch.int32, device=DEV), torch.tensor(rew, device=DEV),
                                      torch.tensor(choices, device=DEV)))
        H = got["t"].size
        dst = (cnt_before + np.arange(H)) % cap
        keep = np.array([j for j in range(H) if not (dst[j + 1:] == dst[j]).any()])
        after = {k: z["c%d_after_%s" % (ci, k)][dst][keep] for k in ("g", "r", "d", "p", "a")}
        assert np.array_equal(got["goal"][keep], after["g"])
        assert np.array_equal(got["reward"][keep], after["r"][:, 0])
        assert np.array_equal(got["done"][keep].astype(np.float32), after["d"][:, 0])
        src = pre + got["t"][keep]
        assert np.array_equal(z["c%d_before_p" % ci][src], after["p"])
        assert np.array_equal(z["c%d_before_a" % ci][src], after["a"])
        assert (cnt_before + H) % cap == cnt_after % cap
        checked += 1
    assert checked >= 3


def test_her_philox_picks_are_distinct_and_reproducible():
    pos, term, trunc, rew = _rollout(4, 512, 128)
    args = [torch.tensor(a, device=DEV) for a in (pos, term, trunc)]
    age0 = torch.zeros(512, dtype=torch.int32, device=DEV)
    r = torch.tensor(rew, device=DEV)
    a = _dev(_ops().her_relabel(*args, age0, r, seed=1))
    b = _dev(_ops().her_relabel(*args, age0, r, seed=1))
    c = _dev(_ops().her_relabel(*args, age0, r, seed=2))
    _assert_same(a, b)
    assert a["t"].size != c["t"].size or not np.array_equal(a["goal"], c["goal"])
    # within an episode the relabelled goals are distinct positions (sampling without replacement)
    ends = np.flatnonzero(a["done"])
    starts = np.concatenate([[0], ends[:-1] + 1])
    seen = {}
    for s, e in zip(starts, ends):
        key = (int(a["n"][e]), int(a["t"][s]))                       # (env, episode start)
        g = tuple(a["goal"][e])
        assert g not in seen.setdefault(key, set())
        seen[key].add(g)
        assert (a["goal"][s:e + 1] == a["goal"][e]).all() and a["t"][e] - a["t"][s] == e - s


# ---------------------------------------------------------------- 9-frame window records (predictor / SoA entry points)
@pytest.mark.parametrize("variant,N,T", [(4, 257, 200), (6, 64, 128)])
@pytest.mark.parametrize("mode", ["choices", "philox"])
def test_window_her_records_match_oracle(variant, N, T, mode):
    """ppo_her_relabel_window with skip = 4 (pre_her_func / pre_f_her_func, env_buffer.py:145-280) vs the oracle, which
    tests/test_window_her_cpu.py pins to the window buffers the reference's own functions produced."""
    pos, term, trunc, rew = _rollout(variant, N, T)
    rs = np.random.RandomState(3 * N + T)
    age0 = np.where(rs.rand(N) < 0.3, rs.randint(1, 20, N), 0).astype(np.int32)
    choices = rs.randint(-1, 12, size=(T, N, 4)).astype(np.int32) if mode == "choices" else None
    want = her_oracle.relabel(pos, term, trunc, age0, rew, choices=choices, seed=SEED, env_id0=9, step0=77, skip=4)
    got = _ops().her_relabel(torch.tensor(pos, device=DEV), torch.tensor(term, device=DEV),
                             torch.tensor(trunc, device=DEV), torch.tensor(age0, device=DEV),
                             torch.tensor(rew, device=DEV),
                             None if choices is None else torch.tensor(choices, device=DEV),
                             seed=SEED, env_id0=9, step0=77, skip=4)
    _assert_same(_dev(got), want)
    assert want["t"].size > 0
    plain = her_oracle.relabel(pos, term, trunc, age0, rew, choices=choices, seed=SEED, env_id0=9, step0=77)
    assert plain["t"].size != want["t"].size or not np.array_equal(plain["goal"], want["goal"])     # skip matters


def test_window_her_against_reference_buffers(golden_dir):
    """The window records the reference's pre_her_func / pre_f_her_func appended (tests/golden/window_her.npz): the device
    kernel's index records (picks replayed), materialised as 9-frame windows from one frame per step, equal them in
    every field."""
    import test_window_her_cpu as wh
    checked = 0
    for ci, meta, names, before, after in wh._cases(golden_dir):
        cap, seed, pre, L, cnt_before, _, cnt_after, _, _, with_f = meta
        if cnt_before <= pre:
            continue
        S, P, A, R, D, LP, F = wh.episode_arrays(before, pre, L, with_f)
        pos = np.ascontiguousarray(P[1:].reshape(L, 1, 2).astype(np.float32))
        term = np.zeros((L, 1), np.uint8); term[-1] = 1
        _, fv = np.unique(before["p"][pre:cnt_before][:, 8, 0:2], return_index=True, axis=0)
        np.random.seed(seed)
        chosen = np.random.choice(fv, size=min(4, fv.size), replace=False)
        choices = np.full((L, 1, 4), -1, np.int32)
        choices[-1, 0, :chosen.size] = [int(np.where(fv == c)[0][0]) for c in chosen]
        got = _dev(_ops().her_relabel(torch.tensor(pos, device=DEV), torch.tensor(term, device=DEV),
                                      torch.zeros((L, 1), dtype=torch.uint8, device=DEV),
                                      torch.zeros(1, dtype=torch.int32, device=DEV),
                                      torch.tensor(R.reshape(L, 1).astype(np.float32), device=DEV),
                                      torch.tensor(choices, device=DEV), skip=4))
        H = got["t"].size
        assert H > 0
        want_w = wh.window_records_from_index_records(got, S, P, A, R, D, LP, F)
        dst = (cnt_before + np.arange(H)) % cap
        keep = np.array([j for j in range(H) if not (dst[j + 1:] == dst[j]).any()])
        for k in names:
            cmp_t = np.int64 if k in ("a", "d") else np.float32
            assert np.array_equal(want_w[k][keep].astype(cmp_t), after[k][dst][keep].astype(cmp_t)), (ci, k)
        assert (cnt_before + H) % cap == cnt_after % cap
        checked += 1
    assert checked >= 4


def test_predictor_trainer_relabels_like_the_window_buffer():
    """VecPPOTrainer.relabel for an agent trained on window records (ppo_predictor.her_window_delay = 4) == the oracle
    with skip = 4 on the trainer's own rollout."""
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.agent.PPO import PPO
    from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    assert getattr(PPO, "her_window_delay", 0) == 0 and ppo_predictor.her_window_delay == 4
    torch.manual_seed(5)
    N, T = 48, 110
    eng = TwoarmyEngine(4, N, 17, seed=SEED)
    tr = VecPPOTrainer(ppo_predictor(), eng, rollout_steps=T, minibatch=512, value_chunk=512)
    tr.collect()
    got = _dev(tr.relabel())
    want = her_oracle.relabel(tr.pos[4:4 + T].cpu().numpy(), tr.term.cpu().numpy(), tr.trunc.cpu().numpy(),
                              np.zeros(N, np.int32), tr.reward.cpu().numpy(), seed=tr.her_seed, env_id0=eng.env_id0,
                              step0=0, skip=4)
    _assert_same(got, want)
    assert want["t"].size > 0
    eng.close()
