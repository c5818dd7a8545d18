"""The arithmetic the fused first-layer kernels implement (include/twoarmy_ppo.h: ppo_conv1_up4_bias_relu / _bwd), in
pure torch on the CPU: a k4/s2 conv over a nearest-x4 upsampled 17x17 image == parity-dependent 2x2-tap convs of the
source image with pre-summed weights (all_net.py:146,176-186), and the gradient map back onto W is its transpose."""
import pytest
import torch


def _ops():
    from twoarmy_amd import ppo_ops
    return ppo_ops


def _folded_forward(x, wf, b):
    """Evaluate the layer from the folded weights exactly as the kernel does: per output parity, four taps."""
    B, F = x.shape[0], x.shape[1]
    xp = torch.nn.functional.pad(x.view(B, F, 17, 17), (0, 1, 0, 1))
    out = torch.zeros(B, 64, 33, 33, dtype=x.dtype)
    for py in (0, 1):
        for px in (0, 1):
            ny, nx = 17 - py, 17 - px
            acc = b.view(1, 64, 1, 1).expand(B, 64, ny, nx).clone()
            for ty in (0, 1):
                for tx in (0, 1):
                    acc += torch.einsum("bchw,co->bohw", xp[:, :, ty:ty + ny, tx:tx + nx], wf[py, px, ty, tx])
            out[:, :, py::2, px::2] = acc
    return out


@pytest.mark.parametrize("F", [4, 8])
def test_folded_weights_reproduce_the_literal_layer(F):
    torch.manual_seed(F)
    w, b = torch.randn(64, F, 4, 4, dtype=torch.float64), torch.randn(64, dtype=torch.float64)
    x = torch.tensor([0.9, -0.9, -0.5, 0.3], dtype=torch.float64)[torch.randint(0, 4, (3, F, 289))]
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.view(3, F, 17, 17), scale_factor=4, mode="nearest"),
                                     w, b, stride=2)
    wf = _ops().fold_conv1_weights(w.float()).double()
    assert wf.shape == (2, 2, 2, 2, F, 64)
    # taps a parity does not have are exactly zero (the kernel reads a zero-padded row / column 17 for them)
    assert float(wf[0, :, 1].abs().max()) == 0.0 and float(wf[:, 0, :, 1].abs().max()) == 0.0
    assert torch.allclose(_folded_forward(x, wf, b), ref, atol=1e-5)


@pytest.mark.parametrize("F", [4, 8])
def test_unfold_is_the_transpose_of_fold(F):
    torch.manual_seed(10 + F)
    ops = _ops()
    w = torch.randn(64, F, 4, 4)
    g = torch.randn(2, 2, 2, 2, F, 64)
    lhs = float((ops.fold_conv1_weights(w).double() * g.double()).sum())
    rhs = float((w.double() * ops.unfold_conv1_grad(g).double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs))
    # and against autograd through the literal layer: d/dW of <conv(up(x), W), gy> == unfold(d/dWf)
    x = torch.randn(2, F, 289, dtype=torch.float64)
    wd = w.double().requires_grad_(True)
    y = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.view(2, F, 17, 17), scale_factor=4, mode="nearest"), wd,
                                   None, stride=2)
    gy = torch.randn_like(y)
    (gw_ref,) = torch.autograd.grad(y, wd, gy)
    xp = torch.nn.functional.pad(x.view(2, F, 17, 17), (0, 1, 0, 1))
    gwf = torch.zeros(2, 2, 2, 2, F, 64, dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            ny, nx = 17 - py, 17 - px
            gsub = gy[:, :, py::2, px::2]
            for ty in (0, 1):
                for tx in (0, 1):
                    gwf[py, px, ty, tx] = torch.einsum("bchw,bohw->co", xp[:, :, ty:ty + ny, tx:tx + nx], gsub)
    assert torch.allclose(ops.unfold_conv1_grad(gwf), gw_ref, atol=1e-8)


def test_decoder_tail_fold_equals_transposed_conv_plus_pooling():
    """ppo_ops.fold_decoder_tail: ConvTranspose2d(16 -> 1, k4, s2) + AvgPool2d(4) of Net_Decoder (all_net.py:100-137) ==
    one 3x3 / stride-2 / pad-1 convolution with the folded taps (what ppo_decoder_frames evaluates from LDS)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from twoarmy_amd import ppo_ops
    torch.manual_seed(3)
    ct = nn.ConvTranspose2d(16, 1, 4, 2)
    a2 = torch.rand(5, 16, 33, 33) - 0.3
    with torch.no_grad():
        want = F.avg_pool2d(ct(a2), 4, 4)
        got = F.conv2d(a2, ppo_ops.fold_decoder_tail(ct.weight).unsqueeze(0), ct.bias, stride=2, padding=1)
    assert want.shape == got.shape == (5, 1, 17, 17)
    assert float((want - got).abs().max()) < 1e-6
