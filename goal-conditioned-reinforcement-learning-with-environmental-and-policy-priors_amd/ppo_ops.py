"""torch front end of the PPO HIP kernels (include/twoarmy_ppo.h).  No CPU fallback: every op
needs device tensors and the HIP library."""
import ctypes as C

import torch

from . import _lib


def _p(t, dtype=None):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "PPO kernels take contiguous device tensors"
    if dtype is not None:
        assert t.dtype == dtype, "expected %s got %s" % (dtype, t.dtype)
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def sample(probs, uniforms=None, seed=0, offset=0, offset_dev=None):
    """Categorical(probs).sample() + log_prob (reference soa/agent/PPO.py:86-88) -> (int32[B], float[B]).
    offset_dev: int64[1] device tensor added to the Philox row counter at run time (graph-captured launches)."""
    B, A = probs.shape
    action = torch.empty(B, dtype=torch.int32, device=probs.device)
    logp = torch.empty(B, dtype=torch.float32, device=probs.device)
    _lib.check(_lib.lib().ppo_sample_dev(_p(probs, torch.float32), B, A, _p(uniforms, torch.float32), seed, offset,
                                         _p(offset_dev, torch.int64), _p(action), _p(logp), _stream(probs)), "ppo_sample")
    return action, logp


def gae(reward, value, next_value, done=None, gamma=0.99, lam=0.0, use_done_mask=False, want_ret=True):
    """Returns (adv, target, ret) [T,N].  lam=0, use_done_mask=False is the reference (PPO.py:113-114)."""
    T, N = reward.shape
    adv = torch.empty_like(reward)
    target = torch.empty_like(reward)
    ret = torch.empty_like(reward) if want_ret else None
    _lib.check(_lib.lib().ppo_gae(_p(reward, torch.float32), _p(value, torch.float32), _p(next_value, torch.float32),
                                  _p(done, torch.uint8), gamma, lam, int(bool(use_done_mask)), T, N, _p(adv), _p(target),
                                  _p(ret), _stream(reward)), "ppo_gae")
    return adv, target, ret


def adv_norm_(adv, eps=1e-8):
    ws = torch.empty(4096, dtype=torch.float64, device=adv.device)
    _lib.check(_lib.lib().ppo_adv_norm(_p(adv, torch.float32), adv.numel(), eps, _p(ws), _stream(adv)), "ppo_adv_norm")
    return adv


class _AttachGrad(torch.autograd.Function):
    """Scalar loss whose gradient w.r.t. `x` was already produced by the fused HIP kernel."""

    @staticmethod
    def forward(ctx, x, loss, grad):
        ctx.save_for_backward(grad)
        ctx.xshape = x.shape
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.xshape), None, None


def ppo_losses(probs, value, action, old_logp, adv, target_v, clip=0.1, ent_coef=0.01, n_valid=None):
    """(action_loss, value_loss) of the reference (PPO.py:124-133).  One fused forward+backward launch;
    the two losses are separate autograd nodes so `action_loss.backward(); value_loss.backward()` works
    exactly like in the reference.  n_valid < B: rows n_valid.. are padding of a fixed-shape minibatch."""
    B, A = probs.shape
    n_valid = B if n_valid is None else int(n_valid)
    with torch.no_grad():
        probs_c = probs.detach().contiguous()
        value_c = value.detach().contiguous().view(-1)
        losses = torch.empty(2, dtype=torch.float32, device=probs.device)
        gp = torch.empty_like(probs_c)
        gv = torch.empty_like(value_c)
        ws = torch.empty(2 * ((B + 255) // 256), dtype=torch.float32, device=probs.device)
        _lib.check(_lib.lib().ppo_loss_fwd_bwd_masked(
            _p(probs_c, torch.float32), _p(action.contiguous(), torch.int32), _p(old_logp.contiguous().view(-1)),
            _p(adv.contiguous().view(-1)), _p(value_c), _p(target_v.contiguous().view(-1)), B, n_valid, A, float(clip),
            float(ent_coef), _p(losses), _p(gp), _p(gv), _p(ws), _stream(probs)), "ppo_loss_fwd_bwd_masked")
    return _AttachGrad.apply(probs, losses[0], gp), _AttachGrad.apply(value, losses[1], gv)


def gather_stack(frames, pos_frames, k_idx, n_idx, age, init_frame, init_pos):
    """frames [K,N,pitch>=289] (may be a [..., :289] view of a 292-pitched buffer) -> ([B,4,289], [B,4,2]).
    uint8 frames are matrix codes (TW_F_MATRIX_CODE) and are expanded to fp32 on the fly."""
    K, N = frames.shape[:2]
    pitch = frames.stride(1)
    assert frames.stride(2) == 1 and frames.stride(0) == N * pitch
    B = k_idx.numel()
    out = torch.empty((B, 4, 289), dtype=torch.float32, device=frames.device)
    pos_out = torch.empty((B, 4, 2), dtype=torch.float32, device=frames.device) if pos_frames is not None else None
    assert frames.dtype in (torch.float32, torch.uint8)
    fn = _lib.lib().ppo_gather_stack if frames.dtype == torch.float32 else _lib.lib().ppo_gather_stack_u8
    _lib.check(fn(
        C.c_void_p(frames.data_ptr()), pitch, _p(pos_frames, torch.float32), N, _p(k_idx, torch.int32),
        _p(n_idx, torch.int32), _p(age, torch.int32), _p(init_frame, torch.float32), _p(init_pos, torch.float32), B,
        _p(out), _p(pos_out), _stream(frames)), "ppo_gather_stack")
    return out, pos_out


def age_scan(terminated, truncated, age0):
    T, N = terminated.shape
    age = torch.empty((T + 1, N), dtype=torch.int32, device=terminated.device)
    _lib.check(_lib.lib().ppo_age_scan(_p(terminated, torch.uint8), _p(truncated, torch.uint8), _p(age0, torch.int32),
                                       T, N, _p(age), _stream(terminated)), "ppo_age_scan")
    return age


def her_relabel(pos, terminated, truncated, age0, reward, choices=None, seed=0, env_id0=0, step0=0, max_goals=4, skip=0):
    """Hindsight relabelling of a time-major rollout (ppo_her_relabel_window, include/twoarmy_ppo.h; reference
    Buffer_gridworld.her_func, soa/env_buffer.py:101-143; skip = 4: pre_her_func / pre_f_her_func on the 9-frame window
    records, :145-280).  Returns a dict of relabelled index records
    {t int32[H], n int32[H], goal f32[H,2], reward f32[H], done u8[H]} and the per-env counts."""
    T, N = terminated.shape
    dev = pos.device
    assert pos.shape == (T, N, 2) and pos.is_contiguous() and reward.shape == (T, N)
    if choices is not None:
        assert choices.shape == (T, N, 4) and choices.dtype == torch.int32 and choices.is_contiguous()
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    args = [_p(pos, torch.float32), _p(terminated.contiguous(), torch.uint8), _p(truncated.contiguous(), torch.uint8),
            _p(age0.contiguous(), torch.int32), _p(reward.contiguous(), torch.float32), _p(choices), int(seed),
            int(env_id0), int(step0), T, N, int(max_goals), int(skip)]
    fn = _lib.lib().ppo_her_relabel_window
    _lib.check(fn(*args, None, _p(counts), None, None, None, None, None, _stream(pos)), "ppo_her_relabel_window")
    incl = torch.cumsum(counts.long(), 0)
    H = int(incl[-1])                                   # the one host sync: the record count sizes the outputs
    offsets = (incl - counts.long()).contiguous()
    out = dict(t=torch.empty(H, dtype=torch.int32, device=dev), n=torch.empty(H, dtype=torch.int32, device=dev),
               goal=torch.empty((H, 2), dtype=torch.float32, device=dev),
               reward=torch.empty(H, dtype=torch.float32, device=dev), done=torch.empty(H, dtype=torch.uint8, device=dev),
               counts=counts)
    if H:
        _lib.check(fn(*args, _p(offsets), _p(counts), _p(out["t"]), _p(out["n"]), _p(out["goal"]), _p(out["reward"]),
                      _p(out["done"]), _stream(pos)), "ppo_her_relabel_window")
    return out


class _ConvBiasReLU(torch.autograd.Function):
    """relu(conv2d(x, w) + b) on channels-last fp32 tensors: the conv (and its two backward GEMMs) in MIOpen, bias +
    ReLU and ReLU-backward + bias-gradient as ONE pass each (ppo_bias_relu_nhwc / ppo_relu_bwd_bias_grad_nhwc) instead
    of the add / clamp / threshold_backward / sum passes PyTorch runs around a Conv2d + ReLU pair."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        y = torch.ops.aten.convolution(x, w, None, list(stride), [0, 0], [1, 1], False, [0, 0], 1)
        assert y.is_contiguous(memory_format=torch.channels_last), "conv epilogue kernels need channels-last activations"
        B, Cc, H, W = y.shape
        _lib.check(_lib.lib().ppo_bias_relu_nhwc(C.c_void_p(y.data_ptr()), _p(b.detach(), torch.float32), B * H * W, Cc,
                                                 _stream(y)), "ppo_bias_relu_nhwc")
        ctx.save_for_backward(x, w, y)
        ctx.stride = list(stride)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        B, Cc, H, W = y.shape
        npix = B * H * W
        blocks = _lib.lib().ppo_relu_bwd_bias_grad_nhwc_blocks(npix, Cc)
        g = torch.empty_like(y)                                   # channels-last like y
        partial = torch.empty((blocks, Cc), dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().ppo_relu_bwd_bias_grad_nhwc(C.c_void_p(gy.data_ptr()), C.c_void_p(y.data_ptr()),
                                                          C.c_void_p(g.data_ptr()), _p(partial), npix, Cc, _stream(y)),
                   "ppo_relu_bwd_bias_grad_nhwc")
        gx, gw, _ = torch.ops.aten.convolution_backward(g, x, w, None, ctx.stride, [0, 0], [1, 1], False, [0, 0], 1,
                                                        [bool(ctx.needs_input_grad[0]), True, False])
        return gx, gw, partial.sum(0), None


def conv_bias_relu(x, weight, bias, stride):
    """relu(conv2d(x, weight, bias, stride)) for channels-last fp32 device tensors (see _ConvBiasReLU)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)
    return _ConvBiasReLU.apply(x, weight, bias, tuple(stride))


def fold_conv1_weights(w):
    """Conv2d(F -> 64, k4, s2) weights [64, F, 4, 4] applied to a nearest-x4 upsampled image == parity-dependent 2x2-tap
    weights on the source image (include/twoarmy_ppo.h): -> float[2][2][2][2][F][64], contiguous."""
    O, F = w.shape[0], w.shape[1]
    w = w.detach().reshape(O, F, 4, 4).float()
    z = torch.zeros_like(w[:, :, 0:1, :])
    rows = torch.stack([torch.cat([w.sum(2, keepdim=True), z], 2),                       # parity 0: all four rows on tap 0
                        torch.cat([w[:, :, 0:2].sum(2, keepdim=True), w[:, :, 2:4].sum(2, keepdim=True)], 2)], 0)  # [py][O][F][ty][4]
    zc = torch.zeros_like(rows[..., 0:1])
    cols = torch.stack([torch.cat([rows.sum(-1, keepdim=True), zc], -1),
                        torch.cat([rows[..., 0:2].sum(-1, keepdim=True), rows[..., 2:4].sum(-1, keepdim=True)], -1)], 1)
    # cols: [py][px][O][F][ty][tx] -> [py][px][ty][tx][F][O]
    return cols.permute(0, 1, 4, 5, 3, 2).contiguous()


@torch.no_grad()
def conv1_up4_bias_relu_infer(frames, weight, bias):
    """Inference-only fused layer for any supported (F, C_out) (include/twoarmy_ppo.h: ppo_conv1_up4_bias_relu_c):
    frames [B, F, 289] -> relu(conv2d(upsample_x4(frames), weight, bias, stride 2)) as a channels-last [B, C, 33, 33]."""
    B, F, _ = frames.shape
    Cout = weight.shape[0]
    y = torch.empty((B, Cout, 33, 33), dtype=torch.float32, device=frames.device, memory_format=torch.channels_last)
    wf = fold_conv1_weights(weight)
    _lib.check(_lib.lib().ppo_conv1_up4_bias_relu_c(_p(frames.contiguous(), torch.float32), B, F, Cout, _p(wf),
                                                    _p(bias.contiguous(), torch.float32), C.c_void_p(y.data_ptr()),
                                                    _stream(y)), "ppo_conv1_up4_bias_relu_c")
    return y


class _Conv1Up4(torch.autograd.Function):
    """relu(conv1(upsample_x4(frames)) + b) in one kernel (ppo_conv1_up4_bias_relu); backward in one kernel too
    (ppo_conv1_up4_bwd: ReLU mask, folded weight gradient and bias gradient from one read of gy and y; the frames
    themselves need no gradient)."""

    @staticmethod
    def forward(ctx, frames, w, b, wf):
        B, F, _ = frames.shape
        y = torch.empty((B, w.shape[0], 33, 33), dtype=torch.float32, device=frames.device,
                        memory_format=torch.channels_last)
        if wf is None:
            wf = fold_conv1_weights(w)
        fr = frames.contiguous()
        _lib.check(_lib.lib().ppo_conv1_up4_bias_relu(_p(fr, torch.float32), B, F, _p(wf), _p(b.detach().contiguous()),
                                                      C.c_void_p(y.data_ptr()), _stream(y)), "ppo_conv1_up4_bias_relu")
        ctx.save_for_backward(fr, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        fr, w, y = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        B, F = fr.shape[0], fr.shape[1]
        groups = _lib.lib().ppo_conv1_up4_bwd_groups(B)
        gw_part = torch.empty((groups, 2, 2, 2, 2, F, 64), dtype=torch.float32, device=y.device)
        gb_part = torch.empty((groups, 4, 64), dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().ppo_conv1_up4_bwd(_p(fr, torch.float32), B, F, C.c_void_p(gy.data_ptr()), C.c_void_p(y.data_ptr()),
                                                _p(gw_part), _p(gb_part), _stream(y)), "ppo_conv1_up4_bwd")
        return None, unfold_conv1_grad(gw_part.sum(0)).to(w.dtype), gb_part.sum((0, 1)), None


def unfold_conv1_grad(gwf):
    """Gradient w.r.t. the folded weights [py][px][ty][tx][F][O] -> gradient w.r.t. W[O][F][4][4] (transpose of
    fold_conv1_weights: row r of W feeds tap 0 of parity 0 and tap r // 2 of parity 1; columns alike)."""
    ty = torch.tensor([[0, 0, 0, 0], [0, 0, 1, 1]], device=gwf.device)              # [py][r]
    out = None
    for py in (0, 1):
        for px in (0, 1):
            g = gwf[py, px][ty[py]][:, ty[px]]                                        # [r][k][F][O]
            out = g if out is None else out + g
    return out.permute(3, 2, 0, 1).contiguous()


def conv1_up4_bias_relu(frames, weight, bias, folded=None):
    """frames [B, F, 289] (F = 4 or 8) -> relu(conv2d(upsample_x4(frames), weight, bias, stride 2)), channels-last.
    folded: fold_conv1_weights(weight) computed by the caller (e.g. once per rollout), or None."""
    assert frames.is_cuda and frames.dtype == torch.float32 and weight.shape[2:] == (4, 4) and weight.shape[0] == 64
    return _Conv1Up4.apply(frames, weight, bias, folded)


def fold_decoder_tail(w3):
    """ConvTranspose2d(16 -> 1, k4, s2) followed by AvgPool2d(4) (Net_Decoder, all_net.py:100-137) as ONE 3x3 / stride-2 /
    pad-1 convolution: pooled cell y averages image rows 4y..4y+3, row 4y+d of the transposed conv reads input row 2y+u
    through tap row d-2u, so input row 2y-1 contributes tap rows {2,3}, row 2y all four, row 2y+1 rows {0,1}."""
    m = torch.tensor([[0., 0., 1., 1.], [1., 1., 1., 1.], [1., 1., 0., 0.]], dtype=w3.dtype, device=w3.device)
    return torch.einsum("ur,crs,vs->cuv", m, w3[:, 0], m).mul_(1.0 / 16.0).contiguous()


def decoder_frames(z, w1, b1, w2, b2, w3, b3):
    """Net_Decoder (inference) in one fused pass per frame: z [n,64,4,4] -> predicted 17x17 frames [n,289]
    (ppo_decoder_frames, include/twoarmy_ppo.h); weights in ConvTranspose2d layout."""
    assert z.dim() == 4 and tuple(z.shape[1:]) == (64, 4, 4) and z.dtype == torch.float32
    z = z.contiguous()
    n = z.shape[0]
    out = torch.empty((n, 289), dtype=torch.float32, device=z.device)
    if n:
        _lib.check(_lib.lib().ppo_decoder_frames(_p(z), n, _p(w1.detach().contiguous(), torch.float32),
                                                 _p(b1.detach().contiguous(), torch.float32),
                                                 _p(w2.detach().contiguous(), torch.float32),
                                                 _p(b2.detach().contiguous(), torch.float32),
                                                 _p(fold_decoder_tail(w3.detach())), 0.0, _p(out), _stream(z)),
                   "ppo_decoder_frames")
        out += b3.detach().view(1, 1)            # the bias as a device-side add: no host synchronisation in the rollout
    return out


def lstm_cell_(gates, c, gates_b=None, bias=None):
    """LSTM cell pointwise ops in one pass (ppo_lstm_cell): pre-activations gates [B,4H] (i, f, g, o; None: zero) + gates_b (optional
    [B,4H] view whose rows may be strided, e.g. xin[:, t] of a [B,T,4H] tensor) + bias (optional [4H]); c [B,H] is
    updated IN PLACE; returns the new hidden state h [B,H]."""
    B, H = c.shape
    H4 = 4 * H
    assert c.dtype == torch.float32 and (gates is not None or gates_b is not None)
    assert gates is None or (gates.shape == (B, H4) and gates.dtype == torch.float32)
    pb, ldb = None, 0
    if gates_b is not None:
        assert gates_b.shape == (B, H4) and gates_b.stride(1) == 1 and gates_b.dtype == torch.float32 and gates_b.is_cuda
        pb, ldb = C.c_void_p(gates_b.data_ptr()), gates_b.stride(0) if B > 1 else H4
    h = torch.empty_like(c)
    _lib.check(_lib.lib().ppo_lstm_cell(_p(gates), pb, ldb, _p(bias), _p(c), _p(h), B, H, _stream(c)), "ppo_lstm_cell")
    return h
