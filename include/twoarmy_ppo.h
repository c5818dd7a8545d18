/*
 * twoarmy_ppo.h -- C ABI of the PPO math kernels (libtwoarmy_hip.so, <package>/csrc/ppo_kernels.hip).
 *
 * Replaces, for batches of B samples resident in HBM, the torch ops of the reference's
 * soa/agent/PPO.py (paths relative to the reference root):
 *   select_action  PPO.py:73-92    Categorical(probs).sample() / log_prob        -> ppo_sample
 *   update         PPO.py:112-115  target_v = r + g*V(s'), adv = target_v - V(s) -> ppo_gae (lambda = 0, no mask)
 *                  PPO.py:115      (commented) adv = (adv - mean) / (std + 1e-8) -> ppo_adv_norm
 *                  PPO.py:124-133  ratio-clip surrogate + entropy, SmoothL1      -> ppo_loss_fwd_bwd
 *   train_ppo.py:116-123  5-frame stack shift + store                            -> ppo_gather_stack
 * GAE(gamma, lambda) with done masks has no reference counterpart: it collapses to the
 * reference formula at lambda = 0, use_done_mask = 0 and is otherwise pinned by oracle/ppo_oracle.py only.
 *
 * Conventions as in twoarmy.h: device pointers, caller-owned, `stream` = hipStream_t as void*,
 * asynchronous, 0 = ok / negative = TW_E_*.  All tensors fp32 unless noted; time-major [T][N].
 */
#ifndef TWOARMY_PPO_H
#define TWOARMY_PPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* torch.distributions.Categorical(probs=p) semantics: q = p / sum(p); logits = log(clamp(q, eps, 1-eps)),
 * eps = FLT_EPSILON; log_prob(a) = logits[a].  Sampling is inverse-CDF on q with a supplied uniform
 * u in [0,1): a = min{k : cumsum(q)[k] > u} (clamped to A-1).  uniforms == NULL -> u from
 * Philox4x32-10(key = seed, counter = (lo32(row + offset), hi32(row + offset), 0, 'TWOS')),
 * u = (word0 >> 8) * 2^-24.
 *   probs float[B][A] (A <= 8), uniforms float[B]|NULL, action int32[B], logp float[B] */
int ppo_sample(const float *probs, int B, int A, const float *uniforms, uint64_t seed, uint64_t offset,
               int32_t *action, float *logp, void *stream);
/* The same with the Philox row counter = row + offset + *offset_dev: a launch recorded in a HIP graph (the whole
 * rollout of VecPPOTrainer is one) takes its position in the stream from device memory at replay time. */
int ppo_sample_dev(const float *probs, int B, int A, const float *uniforms, uint64_t seed, uint64_t offset,
                   const uint64_t *offset_dev, int32_t *action, float *logp, void *stream);

/* delta_t = r_t + gamma * nv_t * cut_t - v_t;  A_t = delta_t + gamma*lambda*cut_t*A_{t+1} (A_T = 0);
 * cut_t = use_done_mask ? 1 - done_t : 1.  Outputs (each nullable): adv = A, target = r + gamma*nv*cut
 * (the reference's target_v), ret = A + v.  Segmented reverse scan: one wavefront scans 64 time steps of
 * one env with 6 shuffle steps over affine maps; [T][N] tiles are transposed through LDS.
 *   reward, value, next_value float[T][N]; done uint8[T][N] (nullable when !use_done_mask) */
int ppo_gae(const float *reward, const float *value, const float *next_value, const uint8_t *done,
            float gamma, float lambda, int use_done_mask, int T, int N, float *adv, float *target, float *ret,
            void *stream);

/* adv <- (adv - mean) / (std + eps), std unbiased (torch.Tensor.std default).  workspace: >= 4096 doubles. */
int ppo_adv_norm(float *adv, int64_t n, float eps, double *workspace, void *stream);

/* Fused forward + backward of the reference losses (PPO.py:124-133) for one minibatch:
 *   action_loss = mean(-min(ratio*adv, clamp(ratio, 1-clip, 1+clip)*adv) - ent_coef * H),
 *   value_loss  = smooth_l1(value, target_v)  (beta = 1, mean)
 * with ratio = exp(logp(a) - old_logp) and Categorical(probs) semantics as in ppo_sample.
 * Writes losses[0] = action_loss, losses[1] = value_loss, d(action_loss)/d(probs) float[B][A] and
 * d(value_loss)/d(value) float[B].  Deterministic (fixed-order two-stage reduction).
 * workspace: >= 2 * ceil(B/256) floats. */
int ppo_loss_fwd_bwd(const float *probs, const int32_t *action, const float *old_logp, const float *adv,
                     const float *value, const float *target_v, int B, int A, float clip, float ent_coef,
                     float *losses, float *grad_probs, float *grad_value, float *workspace, void *stream);
/* The same for a minibatch padded to a fixed shape: rows n_valid .. B-1 are padding (they carry no loss and get zero
 * gradients; the means run over the n_valid real rows).  Keeps every conv launch of an update at ONE batch size:
 * MIOpen searches kernels per shape, and an odd last minibatch costs a new multi-second search. */
int ppo_loss_fwd_bwd_masked(const float *probs, const int32_t *action, const float *old_logp, const float *adv,
                            const float *value, const float *target_v, int B, int n_valid, int A, float clip,
                            float ent_coef, float *losses, float *grad_probs, float *grad_value, float *workspace,
                            void *stream);

/* Policy-input assembly from time-major frames (replaces the 5-deep np.delete/np.append stacks of
 * train_ppo.py:116-121 and the [0:4] / [1:5] slices of PPO.py:113-114,124).  For sample b with newest
 * frame index k_b (row of `frames`), env n_b and age_b = number of env steps taken in the current
 * episode when that newest frame was produced:
 *   out[b][j] = (age_b - (3 - j) <= 0) ? init_frame : frames[k_b - (3 - j)][n_b]      j = 0..3
 * i.e. the four newest frames, never crossing the episode start: older slots repeat the reset frame
 * exactly as np.tile does in Env_transact.reset (env_buffer.py:420-423).  Same for the (y,x) stacks.
 *   frames float[K][N][frame_pitch]; pos_frames float[K][N][2]; k_idx, n_idx, age int32[B];
 *   init_frame float[289]; init_pos float[2]; out float[B][4][289]; pos_out float[B][4][2] (nullable) */
int ppo_gather_stack(const float *frames, int frame_pitch, const float *pos_frames, int N,
                     const int32_t *k_idx, const int32_t *n_idx, const int32_t *age, const float *init_frame,
                     const float *init_pos, int B, float *out, float *pos_out, void *stream);

/* Same, for frames stored as uint8 codes (tw_step/tw_rollout with TW_F_MATRIX_CODE, twoarmy.h): frame_pitch in
 * bytes, each code expanded to its matrix_env value {0: 0.9, 1: -0.9, 2: -0.5, 3: 0.3}; init_frame stays float[289].
 * BASELINE config 5 ("reduced-precision frames"): the stored rollout is 4x smaller and the expansion is exact. */
int ppo_gather_stack_u8(const uint8_t *frames, int frame_pitch, const float *pos_frames, int N,
                        const int32_t *k_idx, const int32_t *n_idx, const int32_t *age, const float *init_frame,
                        const float *init_pos, int B, float *out, float *pos_out, void *stream);

/* Episode age before every step of a rollout: age[0][n] = age0[n]; age[t+1][n] = done[t][n] ? 0 : age[t][n]+1.
 *   done uint8[T][N] (terminated | truncated), age0 int32[N], age int32[T+1][N] */
int ppo_age_scan(const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0, int T, int N,
                 int32_t *age, void *stream);

/* Hindsight experience replay over one time-major rollout: replaces Buffer_gridworld.her_func
 * (soa/env_buffer.py:101-143, called from soa/train_ppo.py:128-134 at every episode end).  For every episode
 * [s0, t1] that starts (age0[n] == 0 or the step after a done) and ends (terminated | truncated) inside the
 * rollout and has <= 64 records:
 *   first_visit = np.unique(achieved (y,x) of its records, axis=0, return_index=True)   (lexicographic order)
 *   k = min(max_goals, len(first_visit)); picks = k entries of first_visit without replacement
 *   for index in picks (in pick order), skipping index == 0:  records 0..index are relabelled with
 *       goal := achieved(index), reward[index] := 0.9, done[index] := 1           (env_buffer.py:120-125)
 * The reference appends copies of those records to its ring buffer; here a relabelled record is the index tuple
 * (t, n, goal, reward, done) -- frames, positions, action and old log-prob are those of (t, n), which the
 * reference copies unchanged.  Output order: env-major, episodes in time order, picks in pick order, prefix in
 * time order (deterministic).
 *   choices int32[T][N][4] | NULL: row (t1, n) holds the picks of the episode ending at t1 as positions in the
 *       sorted unique array (entries outside [0, U) are ignored) -- replays any external RNG, e.g. the global
 *       np.random.choice stream of the reference.  NULL: partial Fisher-Yates with the words of
 *       Philox4x32-10(key = seed, counter = (env_id0 + n, step0 + t1, 0, 'TWOH')), pick j swaps perm[j] with
 *       perm[j + word_j % (U - j)]  (max_goals <= 4).
 * Two passes: offsets == NULL -> only counts[n] (records produced by env n) is written; then, with
 * offsets = exclusive prefix sum of counts (int64[N]), the records are written at offsets[n] ....
 *   pos float[T][N][2] (achieved (y,x) after each step), reward float[T][N], age0 int32[N] (episode age at t=0) */
int ppo_her_relabel(const float *pos, const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0,
                    const float *reward, const int32_t *choices, uint64_t seed, uint32_t env_id0, uint32_t step0, int T,
                    int N, int max_goals, const int64_t *offsets, int32_t *counts, int32_t *out_t, int32_t *out_n,
                    float *out_goal, float *out_reward, uint8_t *out_done, void *stream);
/* The same relabelling for the 9-frame WINDOW records of the predictor / self-orientation entry points
 * (Buffer_gridworld.pre_her_func / pre_f_her_func, soa/env_buffer.py:145-280; windows stored from the fifth step on,
 * train_ppo_predictor.py:134).  Window record i carries the state after step i + 4 as its newest frame, so with
 * skip = 4 the first visits are taken among the states after steps skip, skip + 1, ... only, the first of those is
 * never a goal (`0 < index`), and a pick relabels transitions 0 .. index + skip -- the prefix records plus the four
 * sliding tail windows the reference appends.  skip = 0 is ppo_her_relabel. */
int ppo_her_relabel_window(const float *pos, const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0,
                           const float *reward, const int32_t *choices, uint64_t seed, uint32_t env_id0, uint32_t step0,
                           int T, int N, int max_goals, int skip, const int64_t *offsets, int32_t *counts, int32_t *out_t,
                           int32_t *out_n, float *out_goal, float *out_reward, uint8_t *out_done, void *stream);

/* Epilogues of the conv layers of TINet (all_net.py:141-150: Conv2d + ReLU x 4) on channels-last activations
 * float[n_pixels][C] (n_pixels = B * H * W, C % 4 == 0, C <= 256); the conv GEMMs themselves run in MIOpen.
 *   ppo_bias_relu_nhwc               y <- relu(y + bias[c])  in place: what Conv2d's bias add + nn.ReLU compute, one pass.
 *   ppo_relu_bwd_bias_grad_nhwc      gx = gy * (y > 0)  (ReLU backward on the saved OUTPUT y) and the bias gradient as
 *                                    per-block partial sums partial[blocks][C] (the caller sums them: deterministic);
 *                                    blocks = ppo_relu_bwd_bias_grad_nhwc_blocks(n_pixels, C). */
int ppo_bias_relu_nhwc(float *y, const float *bias, int64_t n_pixels, int C, void *stream);
int ppo_relu_bwd_bias_grad_nhwc_blocks(int64_t n_pixels, int C);
int ppo_relu_bwd_bias_grad_nhwc(const float *gy, const float *y, float *gx, float *partial, int64_t n_pixels, int C,
                                void *stream);

/* First layer of TINet fused with its input upsampling (all_net.py:146,176-186):
 *   out = relu(conv2d(upsample_nearest_x4(frames), W, bias, stride 2))       frames float[B][F][17*17], F = 4 or 8
 * evaluated on the 17x17 frames with parity-folded 2x2-tap weights
 *   folded_w float[2][2][2][2][F][64] = [row parity][column parity][row tap][column tap][in channel][out channel]
 * (row parity 0: tap 0 = W rows 0+1+2+3, tap 1 = 0; parity 1: tap 0 = rows 0+1, tap 1 = rows 2+3; columns alike),
 * out float[B][33][33][64] (channels-last).  Same value as the literal layer up to the summation order. */
int ppo_conv1_up4_bias_relu(const float *frames, int B, int F, const float *folded_w, const float *bias, float *out,
                            void *stream);
/* The same layer shape with C_out output channels: (F, C_out) = (4, 64) / (8, 64) TINet, (1, 16) the world model's
 * Net_Encoder (all_net.py:7-50), whose eval-mode BatchNorm the caller folds into folded_w / bias.
 * folded_w float[2][2][2][2][F][C_out], out float[B][33][33][C_out]. */
int ppo_conv1_up4_bias_relu_c(const float *frames, int B, int F, int C_out, const float *folded_w, const float *bias,
                              float *out, void *stream);

/* Backward of ppo_conv1_up4_bias_relu w.r.t. the folded weights and the bias (the frames carry no gradient):
 *   g = gy * (y > 0);  gw_partial[group][2][2][2][2][F][64] / gb_partial[group][4][64] = per-block partial sums over the
 *   samples a group walks (groups = ppo_conv1_up4_bwd_groups(B); the caller adds groups -- and, for the bias, the four
 *   parities --, then maps the folded gradient back onto W[64][F][4][4]: dW[o][c][r][k] = sum over parities of
 *   gw[py][px][py ? r/2 : 0][px ? k/2 : 0][c][o]).  gy, y: float[B][33][33][64] channels-last, y = the layer's output. */
int ppo_conv1_up4_bwd_groups(int B);
int ppo_conv1_up4_bwd(const float *frames, int B, int F, const float *gy, const float *y, float *gw_partial,
                      float *gb_partial, void *stream);

/* Decoder of the frozen world model, inference only (Net_Decoder, all_net.py:100-137: three ConvTranspose2d with ReLUs
 * between them, then AvgPool2d(4)), one fused pass per frame:
 *   z float[n_frames][64][4][4] -> frames float[n_frames][289]   (the 17x17 predicted state matrix)
 *   w1 float[64][16][2][2], b1[16]; w2 float[16][16][5][5], b2[16]   (ConvTranspose2d layout [C_in][C_out][kH][kW])
 *   kfold float[16][3][3], b3: the last layer (16 -> 1, k4, s2) and the pooling are linear, hence one 3x3 / stride-2 /
 *   pad-1 convolution of the second activation: kfold[c][u][v] = 1/16 * sum of w3[c][0][rows R(u)][columns R(v)],
 *   R(0) = {2, 3}, R(1) = {0, 1, 2, 3}, R(2) = {0, 1}; b3 = the layer's bias.  The 68x68 image is never formed. */
int ppo_decoder_frames(const float *z, int n_frames, const float *w1, const float *b1, const float *w2, const float *b2,
                       const float *kfold, float b3, float *frames, void *stream);

/* Pointwise part of an LSTM cell (nn.LSTM gate order i, f, g, o; the world model's LSTM, all_net.py:52-98, inference):
 *   pre-activations = gates_a float[B][4H] (NULL: absent, e.g. the first step, whose hidden state is zero)  (+ gates_b: rows ldb floats apart, e.g. one time step of the input
 *   projections float[B][T][4H] -> ldb = T * 4H;  NULL: absent)  (+ bias float[4H]; NULL: absent)
 *   c float[B][H] updated in place;   h float[B][H] written
 *   c' = sigmoid(f) c + sigmoid(i) tanh(g),   h' = sigmoid(o) tanh(c').   H % 4 == 0, 16-byte aligned pointers. */
int ppo_lstm_cell(const float *gates_a, const float *gates_b, long long ldb, const float *bias, float *c, float *h, int B,
                  int H, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TWOARMY_PPO_H */
