"""CPU checks of the PPO oracle (oracle/ppo_oracle.py) against values recorded from the reference."""
import numpy as np
import torch

import ppo_oracle as po
from test_ppo_common import det_weights, load_ppo_golden


def test_categorical_logp_entropy_match_reference():
    g = load_ppo_golden()
    q, logits, ent = po.categorical(g["fwd_probs"])
    a = g["buf_a"][:12, 0]
    np.testing.assert_allclose(logits, g["fwd_logits_all"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(logits[np.arange(12), a], g["fwd_logp"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ent, g["fwd_entropy"], rtol=1e-6, atol=1e-7)


def test_first_minibatch_losses_match_reference():
    """oracle losses on the reference's first minibatch (injected weights + recorded permutation)."""
    from twoarmy_amd.soa.agent.net.all_net import Net_PPO_actor, Net_PPO_critic
    g = load_ppo_golden()
    actor, critic = Net_PPO_actor(), Net_PPO_critic()
    actor.load_state_dict(det_weights(actor, 1)); critic.load_state_dict(det_weights(critic, 2))
    s = torch.tensor(g["buf_s"]); p = torch.tensor(g["buf_p"]); goal = torch.tensor(g["buf_g"])
    r = g["buf_r"][:, 0]
    with torch.no_grad():
        nv = critic(s[:, 1:5], p[:, 1:5], goal).numpy()[:, 0]
        v = critic(s[:, 0:4], p[:, 0:4], goal).numpy()[:, 0]
    gamma = float(g["hyper"][0])
    adv, target, _ = po.gae(r[None], v[None], nv[None], np.zeros((1, 64)), gamma, 0.0, False)
    idx = g["upd_perms"][0][:16]
    with torch.no_grad():
        probs = actor(s[idx][:, 0:4], p[idx][:, 0:4], goal[idx]).numpy()
        val = critic(s[idx][:, 0:4], p[idx][:, 0:4], goal[idx]).numpy()[:, 0]
    al, vl = po.losses(probs, g["buf_a"][idx, 0], g["buf_a_logp"][idx, 0], adv[0][idx], val, target[0][idx],
                       clip=float(g["hyper"][1]), ent_coef=float(g["hyper"][2]))
    assert abs(float(al) - g["upd_action_loss"][0]) < 1e-6
    assert abs(float(vl) - g["upd_value_loss"][0]) < 1e-6


def test_gae_lambda0_is_reference_formula_and_general_case_consistent():
    rs = np.random.RandomState(0)
    T, N = 37, 11
    r, v, nv = (rs.randn(T, N).astype(np.float32) for _ in range(3))
    d = (rs.rand(T, N) < 0.1).astype(np.uint8)
    adv, tgt, ret = po.gae(r, v, nv, d, 0.99, 0.0, False)
    assert np.array_equal(tgt, r + np.float32(0.99) * nv) and np.array_equal(adv, tgt - v)
    adv2, _, _ = po.gae(r, v, nv, d, 0.99, 0.95, True)          # float64 closed form
    ref = np.zeros((T, N)); nxt = np.zeros(N)
    for t in range(T - 1, -1, -1):
        cut = 1.0 - d[t]
        nxt = (r[t] + 0.99 * nv[t].astype(np.float64) * cut - v[t]) + 0.99 * 0.95 * cut * nxt
        ref[t] = nxt
    np.testing.assert_allclose(adv2, ref, rtol=2e-5, atol=2e-5)


def test_sample_inverse_cdf():
    probs = np.array([[0.1, 0.2, 0.3, 0.25, 0.15]] * 6, np.float32)
    u = np.array([0.0, 0.0999, 0.1001, 0.59, 0.86, 0.999999], np.float32)
    a, logp = po.sample(probs, u)
    assert a.tolist() == [0, 0, 1, 2, 4, 4]
    np.testing.assert_allclose(logp, np.log(probs[np.arange(6), a]), rtol=1e-6)
