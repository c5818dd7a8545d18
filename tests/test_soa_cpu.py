"""Self-orientation agent vs tests/golden/soa.npz recorded from the reference's own
Self_orientation_agent.py with seeded weights: init, forward of the three heads, update_orientation losses.
Pure torch paths: run on CPU.  (update_policy goes through the HIP loss kernels: tests/test_soa_gpu.py.)"""
import numpy as np
import torch

from test_ppo_common import GOLDEN, det_weights
from test_predictor_cpu import det_weights_v2


def golden():
    return dict(np.load(GOLDEN + "/soa.npz"))


def seeded_agent():
    from twoarmy_amd.soa.agent.Self_orientation_agent import self_orinetation_agent
    agent = self_orinetation_agent()
    for i, net in enumerate((agent.actor, agent.critic, agent.agent_position_preditor)):
        net.load_state_dict(det_weights(net, 21 + i))
    for i, net in enumerate((agent.encoder, agent.decoder)):
        net.load_state_dict(det_weights_v2(net, 13 + i))
    sd = {}
    for k, (name, prm) in enumerate(agent.predictor.state_dict().items()):
        n = prm.numel()
        sd[name] = torch.tensor((0.03 * np.sin(0.37 * np.arange(n, dtype=np.float64) + 1.3 * k)).reshape(tuple(prm.shape)),
                                dtype=prm.dtype)
    agent.predictor.load_state_dict(sd)
    return agent


def buffer_of(g):
    dt = np.dtype([('s', np.float64, (9, 289)), ('a', np.int64, (5, 1)), ('p', np.float64, (9, 2)), ('g', np.float64, (2,)),
                   ('r', np.float64, (5, 1)), ('d', np.int64, (5, 1)), ('a_logp', np.float64, (5, 1)), ('f', np.float64, (5, 2))])
    b = np.zeros(g["buf_s"].shape[0], dtype=dt)
    for k in dt.names:
        b[k] = g["buf_" + k]
    return b


def test_soa_nets_init_like_reference_under_seed():
    g = golden()
    from twoarmy_amd.soa.agent.net.all_net import Net_SoA_actor, Net_SoA_critic, Net_SoA_orient
    torch.manual_seed(9981)
    for tag, net in (("actor", Net_SoA_actor()), ("critic", Net_SoA_critic()), ("orient", Net_SoA_orient())):
        sd = net.state_dict()
        assert list(sd.keys()) == [str(x) for x in g["init_%s_names" % tag]], tag
        assert np.array_equal(np.array([float(v.double().sum()) for v in sd.values()]), g["init_%s_sum" % tag]), tag
        assert np.array_equal(np.array([float(v.double().abs().sum()) for v in sd.values()]), g["init_%s_abs" % tag]), tag


def test_soa_forward_matches_reference():
    g = golden()
    agent = seeded_agent()
    b = buffer_of(g)
    sel = g["fwd_sel"]
    s4 = torch.tensor(b["s"][sel][:, :4], dtype=torch.float32)
    p4 = torch.tensor(b["p"][sel][:, :4], dtype=torch.float32)
    goal = torch.tensor(b["g"][sel], dtype=torch.float32)
    f0 = torch.tensor(b["f"][sel][:, 0], dtype=torch.float32)
    for m in (agent.actor, agent.critic, agent.agent_position_preditor):
        m.eval()
    with torch.no_grad():
        x8 = agent.policy_input(s4)
        px, py = agent.orient_probs(x8, p4, goal)
        cg = torch.cat([goal, f0], 1)
        np.testing.assert_allclose(px.numpy(), g["fwd_px"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(py.numpy(), g["fwd_py"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(agent.actor(x8, p4, cg).numpy(), g["fwd_probs"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(agent.critic(x8, p4, cg).numpy(), g["fwd_value"], rtol=1e-5, atol=1e-5)


def test_update_orientation_losses_match_reference():
    g = golden()
    agent = seeded_agent()
    agent.batch_size_pre_agent, agent.K_epochs_pre_agent_position = 16, 2
    agent.update_orientation(buffer_of(g), "cpu", 0, permutations=g["ori_perms"])
    got = np.array([v for _, v in agent.writer.scalars["loss/future_3steps_loss_update"]])
    assert got.shape == g["ori_loss"].shape
    np.testing.assert_allclose(got, g["ori_loss"], rtol=0, atol=1e-5)
