"""PPO HIP kernels (through the C ABI) vs the numpy oracle, a plain-torch fp32 reference of the same op,
and the update losses recorded from the reference's own PPO.update."""
import numpy as np
import pytest
import torch

import ppo_oracle as po
from test_ppo_common import det_weights, load_ppo_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from twoarmy_amd import ppo_ops
    return ppo_ops


def _rand_probs(rs, B, A=5, extreme=False):
    x = rs.randn(B, A).astype(np.float32) * (8.0 if extreme else 1.5)
    e = np.exp(x - x.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def test_sample_matches_oracle_with_supplied_uniforms():
    rs = np.random.RandomState(1)
    probs = np.concatenate([_rand_probs(rs, 5000), _rand_probs(rs, 3000, extreme=True),
                            np.array([[1, 0, 0, 0, 0], [0, 0, 0, 0, 1], [0.5, 0.5, 0, 0, 0]], np.float32)])
    u = rs.rand(len(probs)).astype(np.float32)
    a, logp = _ops().sample(torch.tensor(probs, device=DEV), torch.tensor(u, device=DEV))
    ra, rl = po.sample(probs, u)
    assert np.array_equal(a.cpu().numpy(), ra)
    np.testing.assert_allclose(logp.cpu().numpy(), rl, rtol=2e-6, atol=2e-6)


def test_sample_philox_stream_is_uniform_and_reproducible():
    probs = torch.full((200000, 5), 0.2, device=DEV)
    a1, _ = _ops().sample(probs, None, seed=9981, offset=0)
    a2, _ = _ops().sample(probs, None, seed=9981, offset=0)
    a3, _ = _ops().sample(probs, None, seed=9981, offset=200000)
    assert torch.equal(a1, a2) and not torch.equal(a1, a3)
    frac = torch.bincount(a1.long(), minlength=5).float() / a1.numel()
    assert float((frac - 0.2).abs().max()) < 0.005


@pytest.mark.parametrize("T,N", [(128, 4096), (1, 2048), (200, 77), (64, 64), (65, 1)])
def test_gae_vs_oracle(T, N):
    rs = np.random.RandomState(T * 1000 + N)
    r, v, nv = (rs.randn(T, N).astype(np.float32) for _ in range(3))
    d = (rs.rand(T, N) < 0.08).astype(np.uint8)
    tr, tv, tnv, td = (torch.tensor(x, device=DEV) for x in (r, v, nv, d))
    # reference mode (PPO.py:113-114): bit-exact
    adv, tgt, ret = _ops().gae(tr, tv, tnv, None, gamma=0.99, lam=0.0, use_done_mask=False)
    oa, ot, orr = po.gae(r, v, nv, d, 0.99, 0.0, False)
    assert np.array_equal(adv.cpu().numpy(), oa) and np.array_equal(tgt.cpu().numpy(), ot)
    assert np.array_equal(ret.cpu().numpy(), orr)
    # GAE(0.99, 0.95) with done masks: parallel scan vs sequential oracle, fp32 tolerance
    adv, tgt, ret = _ops().gae(tr, tv, tnv, td, gamma=0.99, lam=0.95, use_done_mask=True)
    oa, ot, orr = po.gae(r, v, nv, d, 0.99, 0.95, True)
    np.testing.assert_allclose(adv.cpu().numpy(), oa, rtol=1e-5, atol=1e-5)
    assert np.array_equal(tgt.cpu().numpy(), ot)
    np.testing.assert_allclose(ret.cpu().numpy(), orr, rtol=1e-5, atol=1e-5)


def test_adv_norm_vs_oracle():
    rs = np.random.RandomState(3)
    x = (rs.randn(128 * 4096) * 3 + 0.7).astype(np.float32)
    t = torch.tensor(x, device=DEV)
    _ops().adv_norm_(t)
    np.testing.assert_allclose(t.cpu().numpy(), po.adv_norm(x), rtol=1e-5, atol=1e-6)
    ref = torch.tensor(x, device=DEV)
    ref = (ref - ref.mean()) / (ref.std() + 1e-8)                      # the commented line PPO.py:115
    np.testing.assert_allclose(t.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


def _torch_reference_losses(probs, value, a, old_logp, adv, target_v, clip, ent):
    """Plain PyTorch fp32 reference of the same op (the reference's own formulation, PPO.py:124-133)."""
    dist = torch.distributions.Categorical(probs=probs)
    H = dist.entropy().view(-1, 1)
    logp = dist.log_prob(a.long()).view(-1, 1)
    ratio = torch.exp(logp - old_logp.view(-1, 1))
    s1 = ratio * adv.view(-1, 1)
    s2 = torch.clamp(ratio, 1 - clip, 1 + clip) * adv.view(-1, 1)
    al = (-torch.min(s1, s2) - ent * H).mean()
    vl = torch.nn.functional.smooth_l1_loss(value.view(-1, 1), target_v.view(-1, 1))
    return al, vl


@pytest.mark.parametrize("B", [16, 128, 1000, 32768])
def test_loss_fwd_bwd_vs_torch_autograd(B):
    rs = np.random.RandomState(B)
    probs = _rand_probs(rs, B, extreme=(B == 1000))
    if B >= 128:                                     # exact zeros -> clamped logits, zero gradient through the clamp
        probs[:7] = np.array([1, 0, 0, 0, 0], np.float32)
    a = rs.randint(0, 5, B).astype(np.int32)
    old = (np.log(np.maximum(probs[np.arange(B), a], 1e-6)) + rs.randn(B) * 0.15).astype(np.float32)
    adv = rs.randn(B).astype(np.float32)
    adv[:3] = 0.0
    value = rs.randn(B).astype(np.float32) * 2
    target = rs.randn(B).astype(np.float32)
    tp = torch.tensor(probs, device=DEV, requires_grad=True)
    tv = torch.tensor(value, device=DEV, requires_grad=True)
    args = [torch.tensor(x, device=DEV) for x in (a, old, adv, target)]
    al, vl = _ops().ppo_losses(tp, tv.view(-1, 1), args[0], args[1], args[2], args[3], clip=0.1, ent_coef=0.01)
    al.backward(); vl.backward()
    rp = torch.tensor(probs, device=DEV, requires_grad=True)
    rv = torch.tensor(value, device=DEV, requires_grad=True)
    ral, rvl = _torch_reference_losses(rp, rv, args[0], args[1], args[2], args[3], 0.1, 0.01)
    ral.backward(); rvl.backward()
    assert abs(float(al.detach()) - float(ral.detach())) < 1e-5 * max(1.0, abs(float(ral.detach())))        # north_star: 1e-5 fp32
    assert abs(float(vl.detach()) - float(rvl.detach())) < 1e-5 * max(1.0, abs(float(rvl.detach())))
    oal, ovl = po.losses(probs, a, old, adv, value, target)
    assert abs(float(al.detach()) - float(oal)) < 1e-5 and abs(float(vl.detach()) - float(ovl)) < 1e-5
    scale = float(rp.grad.abs().max())
    np.testing.assert_allclose(tp.grad.cpu().numpy(), rp.grad.cpu().numpy(), rtol=2e-4, atol=2e-6 * max(scale, 1e-3))
    np.testing.assert_allclose(tv.grad.cpu().numpy(), rv.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_gather_stack_and_age_scan_vs_oracle():
    rs = np.random.RandomState(5)
    T, N = 40, 33
    term = (rs.rand(T, N) < 0.03).astype(np.uint8)
    trunc = (rs.rand(T, N) < 0.05).astype(np.uint8)
    age0 = rs.randint(0, 6, N).astype(np.int32)
    age = _ops().age_scan(torch.tensor(term, device=DEV), torch.tensor(trunc, device=DEV),
                          torch.tensor(age0, device=DEV)).cpu().numpy()
    ref = np.zeros((T + 1, N), np.int32); ref[0] = age0
    for t in range(T):
        ref[t + 1] = np.where(term[t] | trunc[t], 0, ref[t] + 1)
    assert np.array_equal(age, ref)
    K = T + 4
    buf = torch.tensor(rs.rand(K, N, 292).astype(np.float32), device=DEV)
    frames = buf[..., :289]
    posf = torch.tensor(rs.rand(K, N, 2).astype(np.float32), device=DEV)
    init_f = torch.tensor(rs.rand(289).astype(np.float32), device=DEV)
    init_p = torch.tensor([15.0, 3.0], device=DEV)
    B = 500
    t_idx = rs.randint(0, T, B); n_idx = rs.randint(0, N, B)
    k_idx = (t_idx + 3).astype(np.int32)
    ages = ref[t_idx, n_idx].astype(np.int32)
    out, pos = _ops().gather_stack(frames, posf, torch.tensor(k_idx, device=DEV),
                                   torch.tensor(n_idx.astype(np.int32), device=DEV), torch.tensor(ages, device=DEV),
                                   init_f, init_p)
    ro, rp = po.gather_stack(frames.cpu().numpy(), posf.cpu().numpy(), k_idx, n_idx, ages, init_f.cpu().numpy(),
                             init_p.cpu().numpy())
    assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(pos.cpu().numpy(), rp)


def test_update_matches_reference_losses():
    """PPO.update on the reference's 64-record buffer with injected weights and the recorded minibatch
    permutations: the 8 per-step losses the reference logged must be reproduced within 1e-5 (fp32)."""
    from twoarmy_amd.soa.agent.PPO import PPO
    g = load_ppo_golden()
    agent = PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.batch_size, agent.K_epochs = 16, 2
    buf = np.empty(64, dtype=np.dtype([('s', np.float32, (5, 289)), ('a', np.int64, (1,)), ('p', np.float32, (5, 2)),
                                        ('g', np.float32, (2,)), ('r', np.float32, (1,)), ('d', np.float32, (1,)),
                                        ('a_logp', np.float32, (1,))]))
    for k in buf.dtype.names:
        buf[k] = g["buf_" + k]
    agent.update(buf, DEV, 0, permutations=g["upd_perms"])
    sc = agent.writer.scalars
    al = np.array([v for _, v in sc["loss/action_loss_update"]])
    vl = np.array([v for _, v in sc["loss/value_loss_update"]])
    assert len(al) == 8
    np.testing.assert_allclose(al, g["upd_action_loss"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(vl, g["upd_value_loss"], rtol=0, atol=1e-5)
    for tag, net in (("actor", agent.actor), ("critic", agent.critic)):
        # Adam normalises every element's step to ~lr regardless of gradient magnitude, so elements whose
        # gradient is ~0 can step in opposite directions on different hardware: after 8 steps two correct
        # runs differ by up to 8 * lr per such element.  Checksums are therefore compared per element.
        numel = np.array([p.numel() for p in net.state_dict().values()], np.float64)
        sums = np.array([float(p.detach().double().sum()) for p in net.state_dict().values()])
        abss = np.array([float(p.detach().double().abs().sum()) for p in net.state_dict().values()])
        assert np.all(np.abs(sums - g["upd_%s_sum" % tag]) <= 2e-6 * numel + 1e-4)
        assert np.all(np.abs(abss - g["upd_%s_abs" % tag]) <= 2e-6 * numel + 1e-4)


def test_update_default_randperm_equals_sampler_stream():
    """Without injected permutations the learner draws torch.randperm per epoch like SubsetRandomSampler."""
    from twoarmy_amd.soa.agent.PPO import PPO
    g = load_ppo_golden()
    agent = PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.batch_size, agent.K_epochs = 16, 2
    buf = {k: g["buf_" + k] for k in ("s", "a", "p", "g", "r", "d", "a_logp")}
    torch.manual_seed(123)
    agent.update(buf, DEV, 0)
    al = np.array([v for _, v in agent.writer.scalars["loss/action_loss_update"]])
    np.testing.assert_allclose(al, g["upd_action_loss"], rtol=0, atol=1e-5)


def test_update_matches_reference_losses_channels_last_fused():
    """The fast layout (channels-last conv stacks, MIOpen conv + fused bias/ReLU epilogue kernels, fused ReLU-backward
    + bias-gradient) reproduces the reference's 8 logged update losses within the same 1e-5 as the literal path."""
    from twoarmy_amd.soa.agent.PPO import PPO
    g = load_ppo_golden()
    agent = PPO()
    agent.actor.load_state_dict(det_weights(agent.actor, 1))
    agent.critic.load_state_dict(det_weights(agent.critic, 2))
    agent.batch_size, agent.K_epochs = 16, 2
    agent.to(DEV).use_nhwc()
    buf = {k: g["buf_" + k] for k in ("s", "a", "p", "g", "r", "d", "a_logp")}
    agent.update(buf, DEV, 0, permutations=g["upd_perms"])
    sc = agent.writer.scalars
    np.testing.assert_allclose([v for _, v in sc["loss/action_loss_update"]], g["upd_action_loss"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([v for _, v in sc["loss/value_loss_update"]], g["upd_value_loss"], rtol=0, atol=1e-5)


def test_conv_epilogue_kernels_vs_torch():
    """ppo_bias_relu_nhwc / ppo_relu_bwd_bias_grad_nhwc == relu(conv + b) and its autograd gradients (torch, NCHW)."""
    from twoarmy_amd import ppo_ops
    from twoarmy_amd.soa.agent.net.all_net import use_nhwc
    torch.manual_seed(0)
    for cin, cout, k, hw, B in ((4, 64, 4, 20, 3), (64, 128, 3, 9, 5), (128, 256, 3, 7, 2)):
        conv = torch.nn.Conv2d(cin, cout, k, stride=2).to(DEV)
        x = torch.randn(B, cin, hw, hw, device=DEV, requires_grad=True)
        y_ref = torch.relu(conv(x))
        gy = torch.randn_like(y_ref)
        gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, (x, conv.weight, conv.bias), gy)
        xn = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        use_nhwc([])   # sets the env switch / probes
        wn = conv.weight.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bn = conv.bias.detach().clone().requires_grad_(True)
        y = ppo_ops.conv_bias_relu(xn, wn, bn, (2, 2))
        gx, gw, gb = torch.autograd.grad(y, (xn, wn, bn), gy)
        assert torch.allclose(y, y_ref, atol=1e-5, rtol=1e-5)
        assert torch.allclose(gx, gx_ref, atol=1e-4, rtol=1e-4) and torch.allclose(gw, gw_ref, atol=1e-3, rtol=1e-4)
        assert torch.allclose(gb, gb_ref, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("F", [4, 8])
def test_fused_upsample_conv1_vs_torch(F):
    """ppo_conv1_up4_bias_relu (nearest-x4 upsample folded into conv1, + bias + ReLU) == the literal three modules of
    all_net.py:146,176-186 within fp32 summation-order tolerance, forward and weight / bias gradients."""
    from twoarmy_amd import ppo_ops
    from twoarmy_amd.soa.agent.net.all_net import use_nhwc
    use_nhwc([])
    torch.manual_seed(F)
    B = 37
    conv = torch.nn.Conv2d(F, 64, 4, stride=2).to(DEV)
    frames = torch.tensor([0.9, -0.9, -0.5, 0.3], device=DEV)[torch.randint(0, 4, (B, F, 289), device=DEV)]
    ref = torch.relu(conv(torch.nn.functional.interpolate(frames.view(B, F, 17, 17), scale_factor=4, mode="nearest")))
    gy = torch.randn_like(ref)
    gw_ref, gb_ref = torch.autograd.grad(ref, (conv.weight, conv.bias), gy)
    w = conv.weight.detach().clone().requires_grad_(True)
    b = conv.bias.detach().clone().requires_grad_(True)
    y = ppo_ops.conv1_up4_bias_relu(frames, w, b)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(y, ref, atol=2e-5, rtol=1e-5)
    gw, gb = torch.autograd.grad(y, (w, b), gy)
    # the ReLU mask may differ where |pre-activation| < 1e-5 (summation order); none of these random cases is that close
    assert torch.allclose(gw, gw_ref, atol=2e-3, rtol=1e-4) and torch.allclose(gb, gb_ref, atol=1e-3, rtol=1e-5)


def test_masked_fixed_shape_loss_equals_unpadded():
    """ppo_loss_fwd_bwd_masked: a minibatch padded to a fixed shape (rows n_valid.. are padding) has exactly the losses
    and gradients of the unpadded minibatch; padding rows get zero gradients."""
    from twoarmy_amd import ppo_ops
    torch.manual_seed(5)
    B, P = 300, 212
    probs = torch.softmax(torch.randn(B + P, 5, device=DEV), 1).requires_grad_(True)
    value = torch.randn(B + P, 1, device=DEV, requires_grad=True)
    a = torch.randint(0, 5, (B + P,), device=DEV, dtype=torch.int32)
    old = torch.randn(B + P, 1, device=DEV) * 0.1 - 1.6
    adv, tgt = torch.randn(B + P, 1, device=DEV), torch.randn(B + P, 1, device=DEV)
    la, lv = ppo_ops.ppo_losses(probs, value, a, old, adv, tgt, n_valid=B)
    gp, gv = torch.autograd.grad(la + lv, (probs, value))
    p2, v2 = probs.detach()[:B].clone().requires_grad_(True), value.detach()[:B].clone().requires_grad_(True)
    la2, lv2 = ppo_ops.ppo_losses(p2, v2, a[:B], old[:B], adv[:B], tgt[:B])
    gp2, gv2 = torch.autograd.grad(la2 + lv2, (p2, v2))
    assert float(la) == float(la2) and float(lv) == float(lv2)
    assert torch.equal(gp[:B], gp2) and torch.equal(gv[:B], gv2)
    assert float(gp[B:].abs().max()) == 0.0 and float(gv[B:].abs().max()) == 0.0
