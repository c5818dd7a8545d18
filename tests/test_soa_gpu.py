"""Self-orientation agent on the GPU: update_policy through the HIP target / loss kernels vs the losses the
reference's own update_policy produced (tests/golden/soa.npz), and the batched acting path."""
import numpy as np
import pytest
import torch

from test_soa_cpu import buffer_of, golden, seeded_agent

pytestmark = pytest.mark.gpu


def test_update_policy_losses_match_reference():
    g = golden()
    agent = seeded_agent()
    agent.batch_size, agent.K_epochs = 16, 2
    agent.update_policy(buffer_of(g), "cuda", 0, permutations=g["pol_perms"])
    la = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    lv = np.array([v for _, v in agent.writer.scalars["loss/value_loss_update"]])
    assert la.shape == g["pol_action_loss"].shape
    np.testing.assert_allclose(la, g["pol_action_loss"], rtol=0, atol=1e-5)          # north-star bound: 1e-5 fp32
    np.testing.assert_allclose(lv, g["pol_value_loss"], rtol=0, atol=1e-5)


def test_update_orientation_on_gpu_matches_reference():
    g = golden()
    agent = seeded_agent()
    agent.batch_size_pre_agent, agent.K_epochs_pre_agent_position = 16, 2
    agent.update_orientation(buffer_of(g), "cuda", 0, permutations=g["ori_perms"])
    got = np.array([v for _, v in agent.writer.scalars["loss/future_3steps_loss_update"]])
    np.testing.assert_allclose(got, g["ori_loss"], rtol=0, atol=2e-5)


def test_act_batch_soa_semantics():
    g = golden()
    agent = seeded_agent().to("cuda")
    b = buffer_of(g)
    dev = torch.device("cuda")
    s4 = torch.tensor(b["s"][:, :4], dtype=torch.float32, device=dev)
    p4 = torch.tensor(b["p"][:, :4], dtype=torch.float32, device=dev)
    goal = torch.tensor(b["g"], dtype=torch.float32, device=dev)
    u = torch.rand(s4.shape[0], 3, device=dev)
    a, logp, f = agent.act_batch_soa(s4, p4, goal, u)
    # the same draws by hand: inverse-CDF on the orientation heads, then the policy with the extended goal
    with torch.no_grad():
        x8 = agent.policy_input(s4)
        p0, p1 = agent.orient_probs(x8, p4, goal)
        i0 = (torch.cumsum(p0 / p0.sum(1, keepdim=True), 1) > u[:, 0:1]).float().argmax(1)
        i1 = (torch.cumsum(p1 / p1.sum(1, keepdim=True), 1) > u[:, 1:2]).float().argmax(1)
        probs = agent.actor(x8, p4, torch.cat([goal, f], 1))
    assert torch.equal(f[:, 0], i0.float() - 3) and torch.equal(f[:, 1], i1.float() - 3)
    assert bool(((f >= -3) & (f <= 3)).all()) and bool(((a >= 0) & (a < 5)).all())
    want = torch.log(probs.gather(1, a.long().view(-1, 1)).view(-1))
    assert torch.allclose(logp, want, atol=1e-5)
    act, lp, fx, fy = agent.select_action(b["s"][0][:5], b["p"][0][:5], b["g"][0], dev)
    assert 0 <= act < 5 and -3 <= fx <= 3 and -3 <= fy <= 3 and lp <= 0.0
