"""Actor-critic networks of the PPO path (PyTorch-ROCm: the conv / linear GEMMs run on MFMA via
MIOpen / hipBLASLt; everything around them is HIP, see ppo_ops.py).

Architecture, state_dict key names and the initialisation scheme are those of the reference
(soa/agent/net/all_net.py:139-304) so that checkpoints are interchangeable and, under the same
torch seed, freshly built networks are bit-identical (tests/test_nets.py):

  TINet        frames (B,F,289) -> view (B,F,17,17) -> nearest x4 (B,F,68,68)
               -> conv(F->64,k4,s2) 33^2 -> conv(64->64,k3,s2) 16^2 -> conv(64->128,k4,s2) 7^2
               -> conv(128->256,k3,s2) 3^2 -> flatten 2304 -> fc0 256 ;   all ReLU
               coords (B,4,2)+(B,2) -> Linear(10,128) ; cat 384 -> fc1 512
  actor        bone1 = TINet, A = Linear(512,5), softmax          (keys bone1.*, A.*)
  critic       bone2 = TINet, V = Linear(512,1)                   (keys bone2.*, V.*)
  predictor variants: first conv takes 8 frames (4 real + 4 predicted), all_net.py:249-304.

Init (all_net.py:162-172, applied by TINet and again by the wrapper): Linear xavier-normal / bias 0,
Conv2d xavier-uniform(gain=sqrt 2) / bias 0.1.
"""
import math

import torch
import torch.nn as nn

GRID = 17
CELLS = GRID * GRID


def reference_init(m):
    """The reference's `_weights_init` rule (identical in every network class)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Conv2d):
        nn.init.xavier_uniform_(m.weight, gain=math.sqrt(2.0))
        nn.init.constant_(m.bias, 0.1)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)


def _conv_stack(in_frames):
    spec = [(in_frames, 64, 4), (64, 64, 3), (64, 128, 4), (128, 256, 3)]
    layers = []
    for cin, cout, k in spec:
        layers += [nn.Conv2d(cin, cout, kernel_size=k, stride=2), nn.ReLU()]
    layers.append(nn.Flatten())
    return nn.Sequential(*layers)


class _NHWCFeatures:
    """Marker around the last conv activation (channels-last [B, C, H, W]) on its way to fc0, see TINet._fc0."""

    def __init__(self, t):
        self.t = t


class TINet(nn.Module):
    """Frame-stack + coordinate encoder shared by actor and critic (512-d feature).

    `nhwc` (class switch, set through `use_nhwc`): run the conv stack on channels-last tensors.  MIOpen's MFMA
    implicit-GEMM kernels are NHWC kernels; fed NCHW tensors it wraps every conv in `batched_transpose` launches
    (26 % of a PPO update, profiles/r01_ppo_fp32_kernel_stats_top45.csv) and falls back to a non-MFMA Winograd kernel
    for the stride-2 backward-data convs (another 26 %).  Needs PYTORCH_MIOPEN_SUGGEST_NHWC=1 in the environment
    BEFORE torch is imported (otherwise PyTorch hands MIOpen NCHW copies of the channels-last tensors)."""
    nhwc = False

    def __init__(self):
        super().__init__()
        # construction order == reference order: it fixes both the state_dict order and the
        # torch-RNG consumption of the default initialisers that run before reference_init
        self.cnn_base = _conv_stack(4)
        self.positionnet = nn.Linear(10, 128)
        self.fc0 = nn.Linear(2304, 256)
        self.fc1 = nn.Linear(256 + 128, 512)
        self.upsamplingnearest = nn.UpsamplingNearest2d(scale_factor=4)
        self.apply(reference_init)

    _fold_cache = None

    def _folded_conv1(self, ppo_ops):
        """fold_conv1_weights(conv1.weight), cached across no-grad forwards while the weights are unchanged (the 128
        actor forwards of a rollout, the critic passes of the targets).  The key holds the weight tensor's version
        counter (bumped by every in-place optimiser step) and whether a HIP graph is being captured: a tensor folded
        under capture lives in that graph's memory and is recomputed by its first step at every replay;
        `clear_fold_cache` (called around a capture) keeps it from leaking into another graph or into eager code."""
        if torch.is_grad_enabled():
            return None                                   # training forward: folded inside the op, every call
        w = self.cnn_base[0].weight
        key = (w._version, w.data_ptr(), torch.cuda.is_current_stream_capturing())
        c = self._fold_cache
        if c is None or c[0] != key:
            c = self._fold_cache = (key, ppo_ops.fold_conv1_weights(w))
        return c[1]

    def _convs(self, img, skip_first=False):
        """cnn_base; in channels-last fp32 mode each Conv2d + ReLU pair runs as MIOpen conv + one fused epilogue pass
        (skip_first: `img` already is the first layer's activation)."""
        if not (self.nhwc and img.is_cuda and img.dtype == torch.float32 and not torch.is_autocast_enabled()):
            return self.cnn_base(img)
        from .... import ppo_ops
        x = img
        for k, m in enumerate(self.cnn_base):
            if skip_first and k < 2:
                continue
            if isinstance(m, nn.Conv2d):
                x = ppo_ops.conv_bias_relu(x, m.weight, m.bias, m.stride)
            elif isinstance(m, nn.Flatten):
                x = _NHWCFeatures(x)                      # no copy: fc0 consumes the channels-last order directly
            elif not isinstance(m, nn.ReLU):                  # the ReLUs are part of the fused epilogue
                x = m(x)
        return x

    def _fc0(self, feat):
        """fc0 on the conv features.  Channels-last features (marked by _convs) are consumed as they lie in memory:
        instead of copying B x 2304 floats back into NCHW order (what nn.Flatten does to a channels-last tensor), the
        256 x 2304 weight is viewed in (h, w, c) column order -- a 2.4 MB permute per call instead of B x 9 KB."""
        if isinstance(feat, _NHWCFeatures):
            x = feat.t
            B, C, H, W = x.shape
            w = self.fc0.weight.view(-1, C, H, W).permute(0, 2, 3, 1).reshape(self.fc0.weight.shape[0], -1)
            return torch.nn.functional.linear(x.permute(0, 2, 3, 1).reshape(B, -1), w, self.fc0.bias)
        return self.fc0(feat)

    def widen_input(self, in_frames):
        """Swap the first conv for an `in_frames`-channel one (predictor variants, all_net.py:255,284)."""
        self.cnn_base[0] = nn.Conv2d(in_frames, 64, kernel_size=4, stride=2)

    def forward(self, state_matrix, position, goal):
        B, F, _ = state_matrix.shape
        coords = torch.cat([position.contiguous().view(B, -1), goal], dim=1)
        coords = torch.relu(self.positionnet(coords))
        fused = self.nhwc and state_matrix.is_cuda and state_matrix.dtype == torch.float32 and \
            not torch.is_autocast_enabled() and F in (4, 8)
        if fused:
            # upsample + conv1 + bias + ReLU in one kernel, then MIOpen convs with fused epilogues
            from .... import ppo_ops
            c1 = self.cnn_base[0]
            x = ppo_ops.conv1_up4_bias_relu(state_matrix, c1.weight, c1.bias, self._folded_conv1(ppo_ops))
            feat = torch.relu(self._fc0(self._convs(x, skip_first=True)))
        else:
            img = state_matrix.contiguous().view(B, F, GRID, GRID)
            if self.nhwc:
                img = img.contiguous(memory_format=torch.channels_last)
            img = self.upsamplingnearest(img)
            feat = torch.relu(self._fc0(self._convs(img)))
        return torch.relu(self.fc1(torch.cat([feat, coords], dim=1)))


def clear_fold_cache(modules):
    for m in modules:
        for sub in m.modules():
            if isinstance(sub, TINet):
                sub._fold_cache = None


def use_nhwc(modules, enable=True):
    """Switch the conv stacks of the given networks (already on the GPU) to channels-last weights and activations
    (see TINet.nhwc).  PyTorch reads PYTORCH_MIOPEN_SUGGEST_NHWC once, at the first convolution of the process: it is
    set here if absent, and a probe conv verifies that MIOpen really receives and returns channels-last tensors."""
    import os
    if enable:
        os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
        dev = next(modules[0].parameters()).device if modules else torch.device("cuda", torch.cuda.current_device())
        if dev.type != "cuda":
            raise RuntimeError("use_nhwc: move the networks to the GPU first")
        probe = torch.nn.functional.conv2d(torch.zeros(1, 4, 8, 8, device=dev).contiguous(memory_format=torch.channels_last),
                                           torch.zeros(4, 4, 3, 3, device=dev).contiguous(memory_format=torch.channels_last))
        if not probe.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("channels-last convs need PYTORCH_MIOPEN_SUGGEST_NHWC=1 before the process's first convolution")
    for m in modules:
        for sub in m.modules():
            if isinstance(sub, TINet):
                sub.nhwc = bool(enable)
                sub.cnn_base.to(memory_format=torch.channels_last if enable else torch.contiguous_format)


class _Head(nn.Module):
    bone_name, head_name, head_out, in_frames = "", "", 0, 4
    coord_in = None                 # set: the coordinate MLP is re-created with this many inputs (SoA nets)

    def __init__(self):
        super().__init__()
        bone = TINet()
        if self.in_frames != 4:
            bone.widen_input(self.in_frames)
        if self.coord_in is not None:           # after the conv, like the reference (all_net.py:312-313): RNG order
            bone.positionnet = nn.Linear(self.coord_in, 128)
        setattr(self, self.bone_name, bone)
        for name in ([self.head_name] if isinstance(self.head_name, str) else self.head_name):
            setattr(self, name, nn.Linear(512, self.head_out))
        self.apply(reference_init)

    def features(self, state_matrix, position, goal):
        return getattr(self, self.bone_name)(state_matrix, position, goal)


class Net_PPO_actor(_Head):
    bone_name, head_name, head_out = "bone1", "A", 5

    def forward(self, state_matrix, position, goal, model="actor"):
        return torch.softmax(self.A(self.features(state_matrix, position, goal)), dim=1)


class Net_PPO_critic(_Head):
    bone_name, head_name, head_out = "bone2", "V", 1

    def forward(self, state_matrix, position, goal, model="actor"):
        return self.V(self.features(state_matrix, position, goal))


# ---------------------------------------------------------------------------------------------
# Self-orientation agent (reference all_net.py:306-401): 8 input frames; actor / critic take the goal plus the
# predicted 3-step displacement (4 numbers) next to the 4 (y,x) pairs; the orientation net predicts that displacement
# as two 7-way distributions (offsets -3..3 in y and x).
class Net_SoA_actor(Net_PPO_actor):
    in_frames, coord_in = 8, 12


class Net_SoA_critic(Net_PPO_critic):
    in_frames, coord_in = 8, 12


class Net_SoA_orient(_Head):
    bone_name, head_name, head_out, in_frames, coord_in = "bone3", ("Px", "Py"), 7, 8, 10

    def forward(self, state_matrix, position, goal):
        x = self.features(state_matrix, position, goal)
        return torch.softmax(self.Px(x), dim=1), torch.softmax(self.Py(x), dim=1)


class Net_PPO_Predictor_actor(Net_PPO_actor):
    in_frames = 8


class Net_PPO_Predictor_critic(Net_PPO_critic):
    in_frames = 8


# ---------------------------------------------------------------------------------------------
# Frozen world-model predictor of the PPO+predictor variant (reference all_net.py:7-137): per-frame
# encoder -> 3x1024 LSTM rolled 3 steps past its 4 inputs -> per-frame decoder.  Inference only here
# (PPO_Predictor.py:72-83 runs them under eval() + no_grad); same state_dict keys as the reference.
class Net_Encoder(nn.Module):
    """(B,T,289) -> nearest x4 -> conv(1->16,k4,s2)+BN -> conv(16->16,k5,s4)+BN -> conv(16->64,k2,s2)+BN, ReLU each
    -> latents (B,T,64,4,4) and the upsampled frames (B,T,1,68,68)."""

    def __init__(self):
        super().__init__()
        layers = []
        for cin, cout, k, st in ((1, 16, 4, 2), (16, 16, 5, 4), (16, 64, 2, 2)):
            layers += [nn.Conv2d(cin, cout, kernel_size=k, stride=st), nn.BatchNorm2d(cout), nn.ReLU()]
        self.cnn_base = nn.Sequential(*layers)
        self.apply(reference_init)
        self.upsamplingnearest = nn.UpsamplingNearest2d(scale_factor=4)
        self.device = None                          # kept for attribute compatibility; tensors stay where they are

    def forward(self, state_matrix, need_upsampled=True):
        B, T, _ = state_matrix.shape
        if not need_upsampled and state_matrix.is_cuda and state_matrix.dtype == torch.float32 and \
                not self.training and not torch.is_grad_enabled():
            return self._forward_fused(state_matrix).view(-1, T, 64, 4, 4), None
        up = self.upsamplingnearest(state_matrix.reshape(-1, 1, GRID, GRID)).float()
        z = self.cnn_base(up)
        return z.view(-1, T, 64, 4, 4), up.view(-1, T, 1, 4 * GRID, 4 * GRID)

    def _forward_fused(self, state_matrix):
        """Inference path (eval + no_grad, PPO_Predictor.py:72-83): the first conv sees ONE channel of a x4-upsampled
        17x17 frame -- MIOpen's kernels for that shape took 9.6 ms of the world model's 19.5 ms at 2048 envs.  Here it is
        the parity-folded 2x2-tap conv of the 17x17 frame (ppo_conv1_up4_bias_relu_c, as TINet's first layer) with the
        eval-mode BatchNorm folded into its weights and bias and the ReLU applied before its only store; layers 2-3
        follow on the channels-last activation."""
        from .... import ppo_ops
        conv, bn = self.cnn_base[0], self.cnn_base[1]
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight * scale.view(-1, 1, 1, 1)
        b = (conv.bias - bn.running_mean) * scale + bn.bias
        x = ppo_ops.conv1_up4_bias_relu_infer(state_matrix.reshape(-1, 1, CELLS), w, b)     # [N, 16, 33, 33] channels-last
        # layers 2 (16 -> 16, k5, s4) and 3 (16 -> 64, k2, s2), each with its BatchNorm folded in: MIOpen's kernels for
        # these shapes took ~5 ms per 8192 frames; as a strided-view im2col + ONE GEMM each they are < 1 ms
        N = x.shape[0]
        a = x.permute(0, 2, 3, 1)                                            # [N, 33, 33, 16] view of the same memory
        for conv, bn in ((self.cnn_base[3], self.cnn_base[4]), (self.cnn_base[6], self.cnn_base[7])):
            k, st = conv.kernel_size[0], conv.stride[0]
            Hh, Ci = a.shape[1], a.shape[3]
            Ho = (Hh - k) // st + 1
            sN, sH, sW, sC = a.stride()
            patches = a.as_strided((N, Ho, Ho, k, k, Ci), (sN, st * sH, st * sW, sH, sW, sC)).reshape(N * Ho * Ho, k * k * Ci)
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            wmat = (conv.weight * scale.view(-1, 1, 1, 1)).permute(2, 3, 1, 0).reshape(k * k * Ci, -1)   # [(ky, kx, c), out]
            bvec = (conv.bias - bn.running_mean) * scale + bn.bias
            a = torch.relu(torch.addmm(bvec, patches, wmat)).view(N, Ho, Ho, -1)                      # channels-last again
        return a.permute(0, 3, 1, 2)                                         # [N, 64, 4, 4] (logical NCHW, like the modules)


class LSTM(nn.Module):
    """3-layer LSTM(1024) over the flattened latents; after the T inputs it is fed its own output
    nt-4-1 = 3 more times, returning T+3 latents (reference all_net.py:52-98)."""

    def __init__(self):
        super().__init__()
        self.extrap_t = 4
        self.nt = 8
        self.recurrent_model = nn.LSTM(1024, 1024, num_layers=3, batch_first=True)
        self.device = None

    def forward(self, z_content):
        B, T, D, W, H = z_content.shape
        z_in = z_content.reshape(B, T, D * W * H)
        if z_in.is_cuda and not torch.is_grad_enabled() and not self.training:
            z = self._forward_gemm(z_in)
            return z.reshape(B, self.nt - 1, D, W, H), z_in
        zeros = z_in.new_zeros(3, B, 1024)
        z_past, state = self.recurrent_model(z_in, (zeros, zeros.clone()))
        z_n = z_past[:, -1:].contiguous()
        future = []
        for _ in range(self.nt - 4 - 1):
            z_n, state = self.recurrent_model(z_n, state)
            future.append(z_n)
        z = torch.cat([z_past] + future, dim=1)
        return z.reshape(B, self.nt - 1, D, W, H), z_in

    def _forward_gemm(self, z_in):
        """Inference path of the frozen world model (eval + no_grad, the only way PPO_Predictor.py:72-83 runs it): the
        same LSTM cell arithmetic (i, f, g, o gates; c' = f c + i g; h' = o tanh c') as explicit hipBLASLt GEMMs --
        the input projections of the T known steps of a layer in ONE GEMM (M = B T), the recurrent ones per step --
        plus pointwise ops, instead of MIOpen's RNN (34 TFLOP/s at B = 2048; these 1024 x 4096 GEMMs run ~3x that).
        Then nt - 4 - 1 autoregressive steps through the three layers.  -> [B, T + 3, 1024]"""
        m = self.recurrent_model
        B, T, _ = z_in.shape
        L = m.num_layers
        Wih = [getattr(m, "weight_ih_l%d" % k) for k in range(L)]
        Whh = [getattr(m, "weight_hh_l%d" % k) for k in range(L)]
        bias = [getattr(m, "bias_ih_l%d" % k) + getattr(m, "bias_hh_l%d" % k) for k in range(L)]

        from .... import ppo_ops
        bias = [b.contiguous() for b in bias]
        h = [None] * L                                              # zero initial state: h enters no GEMM at t = 0
        c = [z_in.new_zeros(B, 1024) for _ in range(L)]
        x = z_in                                                    # [B, T, 1024]: the layer's inputs for the known steps
        for k in range(L):
            xin = torch.mm(x.reshape(B * T, -1), Wih[k].t()).view(B, T, -1)                  # one GEMM for all T steps
            outs = []
            for t in range(T):                # h W_hh^T as a plain GEMM (none at t = 0: h is zero); + xin[:, t] (strided
                rec = torch.mm(h[k], Whh[k].t()) if t else None              # view) + bias inside the cell kernel
                h[k] = ppo_ops.lstm_cell_(rec, c[k], xin[:, t], bias[k])
                outs.append(h[k])
            x = torch.stack(outs, dim=1)
        seq = [x]
        z_n = x[:, -1]
        for _ in range(self.nt - 4 - 1):
            inp = z_n
            for k in range(L):
                gates = torch.mm(inp, Wih[k].t())
                gates.addmm_(h[k], Whh[k].t())                      # in place: no copy of the addend
                h[k] = ppo_ops.lstm_cell_(gates, c[k], None, bias[k])
                inp = h[k]
            z_n = inp
            seq.append(z_n.unsqueeze(1))
        return torch.cat(seq, dim=1)


class Net_Decoder(nn.Module):
    """(B,T,64,4,4) -> convT(64->16,k2,s2) -> convT(16->16,k5,s4) -> convT(16->1,k4,s2) = 68x68 -> AvgPool4 -> (B,T,289)."""

    def __init__(self):
        super().__init__()
        self.cnn_base = nn.Sequential(
            nn.ConvTranspose2d(64, 16, kernel_size=2, stride=2), nn.ReLU(),
            nn.ConvTranspose2d(16, 16, kernel_size=5, stride=4), nn.ReLU(),
            nn.ConvTranspose2d(16, 1, kernel_size=4, stride=2))
        self.apply(reference_init)
        self.pool = nn.AvgPool2d(4, stride=4)

    def forward(self, z, need_full=True):
        B, T, D, W, H = z.shape
        if not need_full and z.is_cuda and z.dtype == torch.float32 and not torch.is_grad_enabled() and \
                not torch.is_autocast_enabled():
            # inference of the frozen world model: the three transposed convs + pooling as one fused pass per frame
            # (ppo_decoder_frames); callers that need the 68x68 image (the training losses) take the module path
            from .... import ppo_ops
            c = self.cnn_base
            frames = ppo_ops.decoder_frames(z.reshape(-1, D, W, H), c[0].weight, c[0].bias, c[2].weight, c[2].bias,
                                            c[4].weight, c[4].bias)
            return frames.view(-1, T, CELLS), None
        full = self.cnn_base(z.contiguous().view(-1, D, W, H))
        frames = self.pool(full).view(-1, 1, CELLS).squeeze(-2).reshape(-1, T, CELLS)
        return frames, full.view(-1, T, 1, 4 * GRID, 4 * GRID)
