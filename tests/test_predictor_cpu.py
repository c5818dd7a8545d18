"""Predictor path (frozen encoder -> LSTM -> decoder + 8-frame actor/critic) vs outputs recorded from the
reference's own modules with seeded weights (tests/golden/predictor.npz).  Pure torch: runs on CPU."""
import numpy as np
import torch

from test_ppo_common import GOLDEN, det_weights


def det_weights_v2(module, seed):
    sd = det_weights(module, seed)
    for k, (name, prm) in enumerate(module.state_dict().items()):
        n = prm.numel()
        if name.endswith("num_batches_tracked"):
            sd[name] = prm.clone()
        elif name.endswith("running_var"):
            sd[name] = torch.tensor(1.0 + 0.3 * np.sin(0.37 * np.arange(n) + k + seed), dtype=prm.dtype)
        elif name.endswith("running_mean"):
            sd[name] = torch.tensor(0.1 * np.sin(0.53 * np.arange(n) + k + seed), dtype=prm.dtype)
        elif (".1.weight" in name or ".4.weight" in name or ".7.weight" in name) and prm.dim() == 1:
            sd[name] = torch.tensor(1.0 + 0.2 * np.sin(0.41 * np.arange(n) + k + seed), dtype=prm.dtype)
    return sd


def _agent():
    from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor
    return ppo_predictor


def test_init_and_keys_match_reference_under_seed():
    g = dict(np.load(GOLDEN + "/predictor.npz"))
    torch.manual_seed(9981)
    from twoarmy_amd.soa.agent.net.all_net import (LSTM, Net_Decoder, Net_Encoder, Net_PPO_Predictor_actor,
                                                   Net_PPO_Predictor_critic)
    nets = dict(actor=Net_PPO_Predictor_actor(), critic=Net_PPO_Predictor_critic(), encoder=Net_Encoder(),
                decoder=Net_Decoder())
    lstm = LSTM()
    for tag, net in nets.items():
        sd = net.state_dict()
        assert list(sd.keys()) == [str(x) for x in g["init_%s_names" % tag]], tag
        sums = np.array([float(v.double().sum()) for v in sd.values()])
        abss = np.array([float(v.double().abs().sum()) for v in sd.values()])
        assert np.array_equal(sums, g["init_%s_sum" % tag]) and np.array_equal(abss, g["init_%s_abs" % tag]), tag
    assert list(lstm.state_dict().keys()) == [str(x) for x in g["init_predictor_names"]]
    assert np.array_equal(np.array([float(v.double().sum()) for v in lstm.state_dict().values()]),
                          g["init_predictor_sum"])


def check_pred_states_and_heads(device, tol):
    """pred_states (encoder -> LSTM -> decoder, PPO_Predictor.py:72-83) and the 8-frame actor / critic heads
    (all_net.py:249-304) with injected weights vs the reference's recorded outputs; `tol` scales the tolerances."""
    g = dict(np.load(GOLDEN + "/predictor.npz"))
    agent = _agent()()
    for i, net in enumerate((agent.actor, agent.critic, agent.encoder, agent.decoder)):
        net.load_state_dict(det_weights_v2(net, 11 + i))
    sd = {}
    for k, (name, prm) in enumerate(agent.predictor.state_dict().items()):
        n = prm.numel()
        sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                dtype=prm.dtype)
    agent.predictor.load_state_dict(sd)
    agent.to(device)
    s, p, goal = (torch.tensor(g[k], device=device) for k in ("in_s", "in_p", "in_g"))
    frames, up, full = agent.pred_states(s)
    np.testing.assert_allclose(frames.cpu().numpy(), g["pred_frames"], rtol=1e-5 * tol, atol=1e-6 * tol)
    np.testing.assert_allclose(full.double().sum(dim=(2, 3, 4)).cpu().numpy(), g["pred_full_sum"], rtol=1e-5 * tol,
                               atol=1e-4 * tol)
    x = agent.policy_input(s)
    assert x.shape == (3, 8, 289) and torch.equal(x[:, :4], s)
    agent.actor.eval(); agent.critic.eval()
    with torch.no_grad():
        np.testing.assert_allclose(agent.actor(x, p, goal).cpu().numpy(), g["probs"], rtol=1e-5 * tol, atol=1e-6 * tol)
        np.testing.assert_allclose(agent.critic(x, p, goal).cpu().numpy(), g["value"], rtol=1e-5 * tol, atol=1e-5 * tol)
    return agent, x, p, goal


def test_pred_states_and_heads_match_reference():
    check_pred_states_and_heads("cpu", 1.0)


def test_update_reads_transition_0_of_the_window_records():
    """ppo_predictor._unpack: of a 9-frame window record the update trains on frames 0..4 and slot 0 of a / r / a_logp
    (PPO_Predictor.py:126-155); checked on the window buffer the reference itself updated on (predictor_update.npz)."""
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    g = dict(np.load(GOLDEN + "/predictor_update.npz"))
    dt = Buffer_gridworld.window_dtype(17)
    buf = np.zeros(g["buf_s"].shape[0], dtype=dt)
    for k in dt.names:
        buf[k] = g["buf_" + k]
    s, p, a, goal, r, old = _agent()()._unpack(buf, torch.device("cpu"))
    n = len(buf)
    assert s.shape == (n, 5, 289) and p.shape == (n, 5, 2) and a.shape == (n,) and r.shape == (n,) and old.shape == (n, 1)
    assert np.array_equal(s.numpy(), buf["s"][:, :5].astype(np.float32)) and np.array_equal(p.numpy(), buf["p"][:, :5].astype(np.float32))
    assert np.array_equal(a.numpy(), buf["a"][:, 0, 0]) and np.array_equal(r.numpy(), buf["r"][:, 0, 0].astype(np.float32))
    assert np.array_equal(old.numpy()[:, 0], buf["a_logp"][:, 0, 0].astype(np.float32)) and np.array_equal(goal.numpy(), buf["g"].astype(np.float32))
    assert a.dtype == torch.int32 and int((buf["r"][:, 0, 0] == 0.9).sum()) >= 1          # hindsight records are among them
