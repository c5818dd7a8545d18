// ppo_kernels.hip -- PPO math kernels for MI355X (gfx950), C ABI in include/twoarmy_ppo.h.
//
// Everything here is HBM-bound elementwise / scan work on [T][N] rollout tensors; the conv/linear
// GEMMs of the actor-critic stay in PyTorch-ROCm (MIOpen / hipBLASLt on MFMA).  Reference arithmetic:
// soa/agent/PPO.py:73-92 (select_action), :112-115 (targets / advantages), :124-133 (losses).

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "twoarmy.h"
#include "twoarmy_ppo.h"

namespace {

constexpr float CAT_EPS = FLT_EPSILON;          // torch.finfo(float32).eps used by probs_to_logits

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t &c0, uint32_t &c1, uint32_t &c2,
                                              uint32_t &c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const uint32_t lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// ------------------------------------------------------------------ categorical sample
template <int A>
__global__ void ppo_sample_kernel(const float *__restrict__ probs, int B, const float *__restrict__ uniforms,
                                  uint32_t k0, uint32_t k1, uint64_t offset, const uint64_t *__restrict__ offset_dev,
                                  int32_t *__restrict__ action, float *__restrict__ logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float p[A];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k) { p[k] = probs[(size_t)b * A + k]; sum += p[k]; }
    float u;
    if (uniforms) u = uniforms[b];
    else {
        const uint64_t ctr = (uint64_t)b + offset + (offset_dev ? *offset_dev : 0ull);
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0x54574F53u;   // 'TWOS'
        philox4x32_10(k0, k1, c0, c1, c2, c3);
        u = (float)(c0 >> 8) * (1.0f / 16777216.0f);       // 24 random bits -> [0, 1)
    }
    int a = A - 1;
    float cum = 0.f, qa = 0.f;
    bool found = false;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const float q = p[k] / sum;
        cum += q;
        if (!found && cum > u) { a = k; found = true; }
    }
#pragma unroll
    for (int k = 0; k < A; ++k) if (k == a) qa = p[k] / sum;
    action[b] = a;
    logp[b] = logf(fminf(fmaxf(qa, CAT_EPS), 1.0f - CAT_EPS));
}

// ------------------------------------------------------------------ conv epilogues (channels-last activations)
// The conv GEMMs stay in MIOpen; what PyTorch wraps around each of them -- output.add_(bias) (one pass), ReLU
// (another pass), and in backward threshold_backward (one pass) + the bias-gradient reduction (one pass) -- are pure
// HBM passes over the biggest tensors of the update (33x33x64 floats per sample after conv1).  These two kernels do
// each pair in ONE pass.  Activations are channels-last: element i belongs to channel i % C, C % 4 == 0.
__global__ __launch_bounds__(256) void ppo_bias_relu_kernel(float4 *__restrict__ y, const float4 *__restrict__ bias,
                                                            size_t n4, int c4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = y[i];
        const float4 b = bias[i % c4];
        v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f);
        v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
        y[i] = v;
    }
}

// gx = gy * (y > 0), partial[block][C] = sum over the block's pixels of gx.  Block = 256 threads laid out as
// (256 / c4) pixel rows x c4 channel quads, walking `pixels_per_block` consecutive pixels; the per-thread sums are
// combined through LDS in a fixed order (deterministic), the caller adds the per-block partials.
__global__ __launch_bounds__(256) void ppo_relu_bwd_bias_grad_kernel(const float4 *__restrict__ gy,
                                                                     const float4 *__restrict__ y,
                                                                     float4 *__restrict__ gx,
                                                                     float4 *__restrict__ partial, size_t n_pixels,
                                                                     int c4, int pixels_per_block) {
    __shared__ float4 red[256];
    const int rows = 256 / c4;                        // pixel rows handled side by side (c4 <= 64)
    const int cq = threadIdx.x % c4, row = threadIdx.x / c4;
    const size_t p0 = (size_t)blockIdx.x * pixels_per_block;
    const size_t p1 = p0 + pixels_per_block < n_pixels ? p0 + pixels_per_block : n_pixels;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        for (size_t px = p0 + row; px < p1; px += rows) {
            const size_t i = px * c4 + cq;
            const float4 g = gy[i], v = y[i];
            float4 o;
            o.x = v.x > 0.f ? g.x : 0.f; o.y = v.y > 0.f ? g.y : 0.f;
            o.z = v.z > 0.f ? g.z : 0.f; o.w = v.w > 0.f ? g.w : 0.f;
            gx[i] = o;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (row == 0) {
        for (int r = 1; r < rows; ++r) {
            const float4 o = red[r * c4 + cq];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        partial[(size_t)blockIdx.x * c4 + cq] = acc;
    }
}

// ------------------------------------------------------------------ conv1 of TINet, fused with its input upsampling
// all_net.py:146,176-186: frames (B, F, 17, 17) -> UpsamplingNearest2d(x4) -> Conv2d(F -> 64, k4, s2) -> ReLU = (B, 64, 33, 33).
// Output row oy reads upsampled rows 2oy .. 2oy+3, i.e. source row m = oy >> 1 for all four taps when oy is even and
// source rows m, m+1 (two taps each) when oy is odd; the same for columns.  So the layer is a 2x2-tap convolution
// of the 17x17 frame whose weights depend on the output's (row, column) parity -- 16x fewer input bytes, no 68x68
// tensor, and the bias add and ReLU happen before the one and only store of the layer's (largest) activation.
// Folded weights wf[py][px][ty][tx][c][64] (taps that a parity does not have carry zeros) come from the caller.
// Block = 256 threads = 16 channel quads x 16 pixel slots, one sample per block; per parity phase the thread keeps its
// 4 taps x F x float4 weights in registers and walks the phase's pixels; x values are LDS broadcasts.
template <int F, int CQ>                           // F input channels, CQ = output channels / 4 (16 for TINet, 4 for Net_Encoder)
__global__ __launch_bounds__(256) void ppo_conv1_up4_kernel(const float *__restrict__ frames, const float4 *__restrict__ wf,
                                                            const float4 *__restrict__ bias, float4 *__restrict__ out, int B) {
    __shared__ float xs[F * 18 * 18];                  // [c][18][18]: row / column 17 = zero pad for the (unused) far taps
    constexpr int SLOTS = 256 / CQ;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int cq = tid % CQ, slot = tid / CQ;
    const float *src = frames + (size_t)b * F * 289;
    for (int i = tid; i < F * 324; i += 256) {
        const int c = i / 324, r = i - c * 324, y = r / 18, x = r - y * 18;
        xs[i] = (y < 17 && x < 17) ? src[c * 289 + y * 17 + x] : 0.f;
    }
    __syncthreads();
    const float4 bv = bias[cq];
    float4 *dst = out + (size_t)b * 1089 * CQ;
#pragma unroll 1
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        float4 w[4 * F];
#pragma unroll
        for (int k = 0; k < 4 * F; ++k) w[k] = wf[(ph * 4 * F + k) * CQ + cq];      // [ty][tx][c] -> k = (ty*2+tx)*F + c
        const int ny = 17 - py, nx = 17 - px;               // output rows / columns of this parity
        for (int q = slot; q < ny * nx; q += SLOTS) {
            const int m = q / nx, n = q - m * nx;
            float4 acc = bv;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
#pragma unroll
                for (int c = 0; c < F; ++c) {
                    const float xv = xs[c * 324 + (m + ty) * 18 + n + tx];
                    const float4 ww = w[t * F + c];
                    acc.x = fmaf(xv, ww.x, acc.x); acc.y = fmaf(xv, ww.y, acc.y);
                    acc.z = fmaf(xv, ww.z, acc.z); acc.w = fmaf(xv, ww.w, acc.w);
                }
            }
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
            dst[((2 * m + py) * 33 + 2 * n + px) * CQ + cq] = acc;
        }
    }
}

// Backward of the fused first layer: gw_folded[py][px][ty][tx][c][64] = sum over samples and over the output pixels of
// parity (py, px) of x[c][m+ty][n+tx] * g[pixel][64], g = gy where the layer's output y is positive (ReLU), and the bias
// gradient sum of g -- gy and y are read exactly once, nothing else of the 33x33x64 size is touched (the separate
// ReLU-backward pass + MIOpen weight-gradient conv on re-upsampled frames read and wrote it four times).
// Grid = (groups, 4 parities); a block walks samples blockIdx.x, +groups, ...; block = 16 channel quads x 16 pixel slots,
// 4 taps x F float4 accumulators per thread, combined over the 16 slots through LDS in a fixed order; per-block partial
// sums (deterministic), the caller adds the groups.
template <int F>
__global__ __launch_bounds__(256) void ppo_conv1_up4_bwd_kernel(const float *__restrict__ frames, const float4 *__restrict__ gy,
                                                                const float4 *__restrict__ y, float4 *__restrict__ gw_part,
                                                                float4 *__restrict__ gb_part, int B) {
    __shared__ float xs[F * 18 * 18];
    __shared__ float4 red[256];
    const int tid = threadIdx.x, cq = tid & 15, slot = tid >> 4;
    const int ph = blockIdx.y, py = ph >> 1, px = ph & 1;
    const int ny = 17 - py, nx = 17 - px;
    float4 acc[4 * F];
#pragma unroll
    for (int k = 0; k < 4 * F; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                                    // the previous sample's xs is no longer read
        const float *src = frames + (size_t)b * F * 289;
        for (int i = tid; i < F * 324; i += 256) {
            const int c = i / 324, r = i - c * 324, yy = r / 18, xx = r - yy * 18;
            xs[i] = (yy < 17 && xx < 17) ? src[c * 289 + yy * 17 + xx] : 0.f;
        }
        __syncthreads();
        const size_t base = (size_t)b * 1089 * 16;
        for (int q = slot; q < ny * nx; q += 16) {
            const int m = q / nx, n = q - m * nx;
            const size_t o = base + (size_t)((2 * m + py) * 33 + 2 * n + px) * 16 + cq;
            const float4 gv = gy[o], yv = y[o];
            float4 g;
            g.x = yv.x > 0.f ? gv.x : 0.f; g.y = yv.y > 0.f ? gv.y : 0.f;
            g.z = yv.z > 0.f ? gv.z : 0.f; g.w = yv.w > 0.f ? gv.w : 0.f;
            accb.x += g.x; accb.y += g.y; accb.z += g.z; accb.w += g.w;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
#pragma unroll
                for (int c = 0; c < F; ++c) {
                    const float xv = xs[c * 324 + (m + ty) * 18 + n + tx];
                    float4 &a = acc[t * F + c];
                    a.x = fmaf(xv, g.x, a.x); a.y = fmaf(xv, g.y, a.y); a.z = fmaf(xv, g.z, a.z); a.w = fmaf(xv, g.w, a.w);
                }
            }
        }
    }
    // combine the 16 pixel slots (fixed order), one accumulator at a time
    const size_t wbase = ((size_t)blockIdx.x * 4 + ph) * (4 * F) * 16;
#pragma unroll
    for (int k = 0; k <= 4 * F; ++k) {
        __syncthreads();
        red[tid] = k < 4 * F ? acc[k < 4 * F ? k : 0] : accb;
        __syncthreads();
        if (slot == 0) {
            float4 sum = red[cq];
            for (int r = 1; r < 16; ++r) {
                const float4 o = red[r * 16 + cq];
                sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
            }
            if (k < 4 * F) gw_part[wbase + (size_t)k * 16 + cq] = sum;
            else gb_part[((size_t)blockIdx.x * 4 + ph) * 16 + cq] = sum;
        }
    }
}

// ------------------------------------------------------------------ GAE: segmented reverse scan
// Block = 256 threads = 4 waves, GN envs (columns) x chunks of 64 time steps walked from T backwards (GN = 16 for
// N < 16384 so that a 4096-env rollout still fills 256 workgroups; 64 otherwise).
// Lanes of a wave are TIME within one env column: A_t = d_t + c_t * A_{t+1} is the suffix composition
// of affine maps (c, d), computed with a 6-step Kogge-Stone over __shfl_down; the running A of the
// chunk above enters as the carry.  Tiles go through LDS so that global accesses stay coalesced.
constexpr int GT = 64;     // time steps per chunk

template <int GN>          // envs per block
__global__ __launch_bounds__(256) void ppo_gae_kernel(const float *__restrict__ reward, const float *__restrict__ value,
                                                      const float *__restrict__ next_value,
                                                      const uint8_t *__restrict__ done, float gamma, float lambda,
                                                      int use_done_mask, int T, int N, float *__restrict__ adv,
                                                      float *__restrict__ target, float *__restrict__ ret) {
    __shared__ float sd[GT][GN + 1];
    __shared__ float sc[GT][GN + 1];
    __shared__ float carry[GN];
    constexpr int ROWS = 256 / GN;                      // time rows loaded / stored per pass
    const int tid = threadIdx.x, tx = tid % GN, ty = tid / GN;
    const int n = blockIdx.x * GN + tx;
    if (tid < GN) carry[tid] = 0.f;
    const int nchunks = (T + GT - 1) / GT;
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        const int t0 = ch * GT;
        __syncthreads();
        for (int r = ty; r < GT; r += ROWS) {
            const int t = t0 + r;
            float d = 0.f, c = 1.f;                     // identity map for rows past T
            if (t < T && n < N) {
                const size_t i = (size_t)t * N + n;
                const float cut = (use_done_mask && done[i]) ? 0.f : 1.f;
                // same rounding sequence as torch: g*V, * cut, + r, - V.  cut is 0 or 1, so a contracted
                // fma(cut, g*V, r) rounds exactly like the separate multiply and add.
                const float tv = reward[i] + (gamma * next_value[i]) * cut;
                if (target) target[i] = tv;
                d = tv - value[i];
                c = gamma * lambda * cut;
            }
            sd[r][tx] = d; sc[r][tx] = c;
        }
        __syncthreads();
        const int wave = tid >> 6, lane = tid & 63;     // lane = time row inside the chunk
        for (int e = wave * (GN / 4); e < (wave + 1) * (GN / 4); ++e) {
            float d = sd[lane][e], c = sc[lane][e];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float d2 = __shfl_down(d, off), c2 = __shfl_down(c, off);
                if (lane + off < 64) { d = fmaf(c, d2, d); c = c * c2; }
            }
            const float a = fmaf(c, carry[e], d);       // A of this row given the chunk above
            sd[lane][e] = a;
            const float a0 = __shfl(a, 0);
            if (lane == 0) carry[e] = a0;               // only this wave touches carry[e]
        }
        __syncthreads();
        for (int r = ty; r < GT; r += ROWS) {
            const int t = t0 + r;
            if (t < T && n < N) {
                const size_t i = (size_t)t * N + n;
                const float a = sd[r][tx];
                if (adv) adv[i] = a;
                if (ret) ret[i] = a + value[i];
            }
        }
    }
}

// ------------------------------------------------------------------ advantage normalisation
__global__ __launch_bounds__(256) void ppo_moments_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ ws) {
    __shared__ double s1[256], s2[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i];
        a += v; b += v * v;
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ws[2 * blockIdx.x] = s1[0]; ws[2 * blockIdx.x + 1] = s2[0]; }
}

__global__ __launch_bounds__(256) void ppo_normalise_kernel(float *__restrict__ x, int64_t n, float eps,
                                                            const double *__restrict__ ws, int nblocks) {
    __shared__ float s_mean, s_inv;
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = (int)threadIdx.x < nblocks ? ws[2 * threadIdx.x] : 0.0;      // nblocks <= 256 partial sums
    r2[threadIdx.x] = (int)threadIdx.x < nblocks ? ws[2 * threadIdx.x + 1] : 0.0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {                                            // fixed-order tree: deterministic
        if ((int)threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double a = r1[0], b = r2[0];
        const double mean = a / (double)n;
        double var = n > 1 ? (b - (double)n * mean * mean) / (double)(n - 1) : 0.0;  // unbiased
        if (var < 0.0) var = 0.0;
        s_mean = (float)mean;
        s_inv = 1.0f / ((float)sqrt(var) + eps);
    }
    __syncthreads();
    const float mean = s_mean, inv = s_inv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = (x[i] - mean) * inv;
}

// ------------------------------------------------------------------ fused PPO losses, forward + backward
template <int A>
__global__ __launch_bounds__(256) void ppo_loss_kernel(const float *__restrict__ probs, const int32_t *__restrict__ action,
                                                       const float *__restrict__ old_logp, const float *__restrict__ adv,
                                                       const float *__restrict__ value, const float *__restrict__ target_v,
                                                       int B, int n_valid, float clip, float ent_coef,
                                                       float *__restrict__ grad_probs, float *__restrict__ grad_value,
                                                       float *__restrict__ ws) {
    __shared__ float sa[256], sv[256];
    const int b = blockIdx.x * 256 + threadIdx.x;
    float la = 0.f, lv = 0.f;
    if (b >= n_valid && b < B) {                      // padding rows of a fixed-shape minibatch: no loss, no gradient
#pragma unroll
        for (int k = 0; k < A; ++k) grad_probs[(size_t)b * A + k] = 0.f;
        grad_value[b] = 0.f;
    }
    if (b < n_valid) {
        const float invB = 1.0f / (float)n_valid;
        float p[A], q[A], l[A];
        bool inside[A];
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) { p[k] = probs[(size_t)b * A + k]; S += p[k]; }
        float H = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            q[k] = p[k] / S;
            inside[k] = q[k] >= CAT_EPS && q[k] <= 1.0f - CAT_EPS;        // clamp passes gradient inside (inclusive)
            l[k] = logf(fminf(fmaxf(q[k], CAT_EPS), 1.0f - CAT_EPS));
            H -= q[k] * l[k];
        }
        const int a = action[b];
        float la_logp = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) if (k == a) la_logp = l[k];
        const float ratio = expf(la_logp - old_logp[b]);
        const float ad = adv[b];
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float s1 = ratio * ad;
        const float s2 = fminf(fmaxf(ratio, lo), hi) * ad;
        la = -fminf(s1, s2) - ent_coef * H;
        // d/d(ratio) of min(s1, s2): torch.minimum splits ties; clamp passes gradient on [lo, hi]
        const bool in_clip = ratio >= lo && ratio <= hi;
        float w1 = s1 < s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
        float w2 = s2 < s1 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
        const float g_ratio = -(w1 + (in_clip ? w2 : 0.f)) * ad;
        const float g_logp = g_ratio * ratio;
        // gradient w.r.t. normalised q, then through q = p / S
        float gq[A];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            float g = (k == a && inside[k]) ? g_logp / q[k] : 0.f;
            g += ent_coef * (l[k] + (inside[k] ? 1.f : 0.f));            // -ent_coef * dH/dq_k
            gq[k] = g;
            dot += g * q[k];
        }
#pragma unroll
        for (int k = 0; k < A; ++k) grad_probs[(size_t)b * A + k] = (gq[k] - dot) / S * invB;
        // SmoothL1 (beta = 1)
        const float d = value[b] - target_v[b];
        const float ab = fabsf(d);
        lv = ab < 1.0f ? 0.5f * d * d : ab - 0.5f;
        grad_value[b] = (ab < 1.0f ? d : (d > 0.f ? 1.f : -1.f)) * invB;
    }
    sa[threadIdx.x] = la; sv[threadIdx.x] = lv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sv[threadIdx.x] += sv[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ws[2 * blockIdx.x] = sa[0]; ws[2 * blockIdx.x + 1] = sv[0]; }
}

__global__ void ppo_loss_finalize_kernel(const float *__restrict__ ws, int nblocks, int B, float *__restrict__ losses) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float a = 0.f, v = 0.f;
        for (int k = 0; k < nblocks; ++k) { a += ws[2 * k]; v += ws[2 * k + 1]; }       // fixed order
        losses[0] = a / (float)B;
        losses[1] = v / (float)B;
    }
}

// ------------------------------------------------------------------ stack gather / age scan
// frame element -> fp32 policy input: identity for float frames, 4-entry LUT for uint8 code frames (TW_F_MATRIX_CODE)
__device__ __forceinline__ float frame_value(float v) { return v; }
__device__ __forceinline__ float frame_value(uint8_t c) {
    return c == 0 ? 0.9f : (c == 1 ? -0.9f : (c == 2 ? -0.5f : 0.3f));
}

// One wavefront per SAMPLE (block = 4 waves = 4 samples): the three index words are fetched once, then the loads of
// all four stack slots (4 x 289 elements) are issued before any of them is stored -- 4.6 KB in flight per wave instead
// of the 1.2 KB of the wave-per-(sample, slot) version, which sat at 0.42 of the HBM peak on its dependent
// index -> row round trips.
template <typename FT>
__global__ __launch_bounds__(256) void ppo_gather_stack_kernel(const FT *__restrict__ frames, int frame_pitch,
                                                               const float *__restrict__ pos_frames, int N,
                                                               const int32_t *__restrict__ k_idx,
                                                               const int32_t *__restrict__ n_idx,
                                                               const int32_t *__restrict__ age,
                                                               const float *__restrict__ init_frame,
                                                               const float *__restrict__ init_pos, int B,
                                                               float *__restrict__ out, float *__restrict__ pos_out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int k = k_idx[b], n = n_idx[b], ag = age[b];
    constexpr int PER = (TW_CELLS + 63) / 64;                  // 5 elements per lane and slot
    float v[4][PER];
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int back = 3 - j;
        const bool use_init = ag - back <= 0;
        const size_t row = ((size_t)(k - back) * N + n);
        const FT *src = frames + row * frame_pitch;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            v[j][i] = c < TW_CELLS ? (use_init ? init_frame[c] : frame_value(src[c])) : 0.f;
        }
        if (pos_out && lane < 2) ps[j] = use_init ? init_pos[lane] : pos_frames[row * 2 + lane];
    }
    float *dst = out + (size_t)b * 4 * TW_CELLS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            if (c < TW_CELLS) dst[j * TW_CELLS + c] = v[j][i];
        }
        if (pos_out && lane < 2) pos_out[((size_t)b * 4 + j) * 2 + lane] = ps[j];
    }
}

// ------------------------------------------------------------------ hindsight relabelling (HER)
// Buffer_gridworld.her_func (soa/env_buffer.py:101-143) over a time-major rollout.  The reference copies
// episode prefixes into its ring buffer; here a relabelled transition is only an index record
// (t, n, goal', reward', done'): the frames / positions / action / old log-prob are those of (t, n).
// One wavefront per env walks its episodes in time order; an episode (<= 64 steps) sits one step per lane.
constexpr int HER_MAX_LEN = 64;

__global__ __launch_bounds__(64) void ppo_her_kernel(const float *__restrict__ pos, const uint8_t *__restrict__ terminated,
                                                     const uint8_t *__restrict__ truncated,
                                                     const int32_t *__restrict__ age0, const float *__restrict__ reward,
                                                     const int32_t *__restrict__ choices, uint32_t k0, uint32_t k1,
                                                     uint32_t env_id0, uint32_t step0, int T, int N, int max_goals, int skip,
                                                     const int64_t *__restrict__ offsets, int32_t *__restrict__ counts,
                                                     int32_t *__restrict__ out_t, int32_t *__restrict__ out_n,
                                                     float *__restrict__ out_goal, float *__restrict__ out_reward,
                                                     uint8_t *__restrict__ out_done) {
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    if (n >= N) return;
    const bool emit = offsets != nullptr;
    int64_t base = emit ? offsets[n] : 0;
    int total = 0;
    int start = age0[n] == 0 ? 0 : -1;               // an episode already running at t = 0 is not relabelled
    for (int c0 = 0; c0 < T; c0 += 64) {
        const int tl = c0 + lane;
        const bool d = tl < T && (terminated[(size_t)tl * N + n] | truncated[(size_t)tl * N + n]) != 0;
        unsigned long long dm = __ballot(d);
        while (dm) {
            const int t1 = c0 + __builtin_ctzll(dm);
            dm &= dm - 1;
            const int s0 = start;
            start = t1 + 1;
            if (s0 < 0) continue;
            const int L = t1 - s0 + 1;
            if (L > HER_MAX_LEN) continue;
            // lane i = record i of the episode: achieved (y, x) after step s0 + i
            const bool in = lane < L;
            const size_t row = (size_t)(s0 + (in ? lane : 0)) * N + n;
            const float py = pos[row * 2], px = pos[row * 2 + 1];
            // np.unique(axis=0): first occurrence of every distinct (y, x), ordered lexicographically by (y, x).
            // Window records (skip = 4, env_buffer.py:145-280) only offer the states after steps skip, skip+1, ...
            bool first = in && lane >= skip;
            for (int j = skip; j < L; ++j) {
                const float qy = __shfl(py, j), qx = __shfl(px, j);
                if (j < lane && qy == py && qx == px) first = false;
            }
            const unsigned long long fm = __ballot(first);
            const int U = __popcll(fm);
            int rank = 0;
            for (int j = 0; j < L; ++j) {
                const float qy = __shfl(py, j), qx = __shfl(px, j);
                if (((fm >> j) & 1ull) && (qy < py || (qy == py && qx < px))) ++rank;
            }
            const int k = max_goals < U ? max_goals : U;
            // Fisher-Yates over the U unique entries when no explicit picks are supplied (lane j holds perm[j])
            int perm = lane;
            if (!choices) {
                uint32_t w0 = env_id0 + (uint32_t)n, w1 = step0 + (uint32_t)t1, w2 = 0, w3 = 0x54574F48u;   // 'TWOH'
                philox4x32_10(k0, k1, w0, w1, w2, w3);
                const uint32_t w[4] = {w0, w1, w2, w3};
                for (int j = 0; j < k && j < 4; ++j) {
                    const int sidx = j + (int)(w[j] % (uint32_t)(U - j));
                    const int pj = __shfl(perm, j), ps = __shfl(perm, sidx);
                    if (lane == j) perm = ps;
                    else if (lane == sidx) perm = pj;
                }
            }
            for (int j = 0; j < k; ++j) {
                const int pick = choices ? choices[((size_t)t1 * N + n) * 4 + j] : __shfl(perm, j);
                if (pick < 0 || pick >= U) continue;
                const unsigned long long hit = __ballot(first && rank == pick);
                if (!hit) continue;
                const int idx = __builtin_ctzll(hit);
                if (idx == skip) continue;                               // env_buffer.py:119 / :163 `if 0 < index < cap`
                if (emit && lane <= idx) {
                    const int64_t o = base + lane;
                    out_t[o] = s0 + lane;
                    out_n[o] = n;
                    out_goal[o * 2] = __shfl(py, idx);
                    out_goal[o * 2 + 1] = __shfl(px, idx);
                    out_reward[o] = lane == idx ? 0.9f : reward[row];
                    out_done[o] = lane == idx ? 1 : 0;
                }
                base += idx + 1;
                total += idx + 1;
            }
        }
    }
    if (lane == 0 && counts) counts[n] = total;
}

__global__ void ppo_age_scan_kernel(const uint8_t *__restrict__ terminated, const uint8_t *__restrict__ truncated,
                                    const int32_t *__restrict__ age0, int T, int N, int32_t *__restrict__ age) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int a = age0[n];
    age[n] = a;
    for (int t = 0; t < T; ++t) {
        const size_t i = (size_t)t * N + n;
        a = (terminated[i] | truncated[i]) ? 0 : a + 1;
        age[(size_t)(t + 1) * N + n] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// World-model decoder, inference only (Net_Decoder, all_net.py:100-137): latent float[64][4][4] per frame ->
//   ConvTranspose2d(64->16, k2, s2) + ReLU  -> a1[8][8][16]
//   ConvTranspose2d(16->16, k5, s4) + ReLU  -> a2[33][33][16]
//   ConvTranspose2d(16->1,  k4, s2)         -> 68x68,  AvgPool2d(4) -> 17x17
// The last layer and the pooling are both linear, so together they are ONE 3x3 / stride-2 / pad-1 convolution of a2
// with pre-summed taps (kfold, see ppo_decoder_frames in twoarmy_ppo.h): the 68x68 image never exists.  One workgroup
// walks frames with a grid stride; weights stay in LDS, a frame's activations never leave LDS (120 KB per workgroup).
constexpr int DEC_A2 = 33 * 33 * 16;

__global__ __launch_bounds__(256) void ppo_decoder_frames_kernel(const float *__restrict__ z, int n_frames,
                                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                                 const float *__restrict__ w2, const float *__restrict__ b2,
                                                                 const float *__restrict__ kfold, float b3,
                                                                 float *__restrict__ frames) {
    __shared__ __attribute__((aligned(16))) float W2s[16 * 25 * 16];      // [ci][tap][co]
    __shared__ __attribute__((aligned(16))) float W1s[64 * 4 * 16];       // [ci][ky*2+kx][co]
    __shared__ __attribute__((aligned(16))) float Ks[9 * 16];             // [u*3+v][c]
    __shared__ __attribute__((aligned(16))) float B1s[16], B2s[16];
    __shared__ __attribute__((aligned(16))) float zs[64 * 16];            // [ci][iy*4+ix]
    __shared__ __attribute__((aligned(16))) float a1s[64 * 16];           // [oy*8+ox][c]
    __shared__ __attribute__((aligned(16))) float a2s[DEC_A2];            // [oy*33+ox][c]
    const int tid = threadIdx.x;
    // ConvTranspose2d weights are [C_in][C_out][kH][kW]
    for (int i = tid; i < 16 * 16 * 25; i += 256) {
        const int ci = i / 400, r = i - ci * 400, co = r / 25, tap = r - co * 25;
        W2s[(ci * 25 + tap) * 16 + co] = w2[i];
    }
    for (int i = tid; i < 64 * 16 * 4; i += 256) {
        const int ci = i >> 6, r = i & 63, co = r >> 2, k = r & 3;
        W1s[(ci * 4 + k) * 16 + co] = w1[i];
    }
    if (tid < 144) { const int c = tid / 9, t = tid - c * 9; Ks[t * 16 + c] = kfold[tid]; }     // kfold is [c][u][v]
    if (tid < 16) { B1s[tid] = b1[tid]; B2s[tid] = b2[tid]; }
    for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
        __syncthreads();                                   // weights loaded / previous frame's stage 3 done with a2s
        for (int i = tid; i < 1024; i += 256) zs[i] = z[(size_t)f * 1024 + i];
        __syncthreads();
        // ---- stage 1: every output pixel has exactly one source pixel (k2, s2)
        for (int o = tid; o < 1024; o += 256) {
            const int co = o & 15, pix = o >> 4, oy = pix >> 3, ox = pix & 7;
            const int src = (oy >> 1) * 4 + (ox >> 1), k = (oy & 1) * 2 + (ox & 1);
            float acc = B1s[co];
#pragma unroll 8
            for (int ci = 0; ci < 64; ++ci) acc = fmaf(zs[ci * 16 + src], W1s[(ci * 4 + k) * 16 + co], acc);
            a1s[pix * 16 + co] = fmaxf(acc, 0.0f);
        }
        __syncthreads();
        // ---- stage 2: output row oy receives input row oy>>2 through tap row oy&3 and, where oy is a multiple of 4,
        //      input row (oy>>2)-1 through tap row 4 (k5, s4: neighbouring patches overlap by one line); columns alike
        for (int item = tid; item < 33 * 33 * 4; item += 256) {
            const int q = item & 3, pix = item >> 2, oy = pix / 33, ox = pix - oy * 33;
            float4 acc = *reinterpret_cast<const float4 *>(&B2s[q * 4]);
            const int iy0 = oy >> 2, ky0 = oy & 3, ix0 = ox >> 2, kx0 = ox & 3;
#pragma unroll
            for (int ry = 0; ry < 2; ++ry) {
                const int iy = ry ? iy0 - 1 : iy0, ky = ry ? 4 : ky0;
                if (ry ? (ky0 != 0 || iy0 == 0) : (iy0 > 7)) continue;
#pragma unroll
                for (int rx = 0; rx < 2; ++rx) {
                    const int ix = rx ? ix0 - 1 : ix0, kx = rx ? 4 : kx0;
                    if (rx ? (kx0 != 0 || ix0 == 0) : (ix0 > 7)) continue;
                    const float *a = &a1s[(iy * 8 + ix) * 16];
                    const float *w = &W2s[(ky * 5 + kx) * 16 + q * 4];
#pragma unroll
                    for (int ci = 0; ci < 16; ++ci) {
                        const float av = a[ci];
                        const float4 wv = *reinterpret_cast<const float4 *>(w + ci * 400);
                        acc.x = fmaf(av, wv.x, acc.x); acc.y = fmaf(av, wv.y, acc.y);
                        acc.z = fmaf(av, wv.z, acc.z); acc.w = fmaf(av, wv.w, acc.w);
                    }
                }
            }
            acc.x = fmaxf(acc.x, 0.0f); acc.y = fmaxf(acc.y, 0.0f); acc.z = fmaxf(acc.z, 0.0f); acc.w = fmaxf(acc.w, 0.0f);
            *reinterpret_cast<float4 *>(&a2s[pix * 16 + q * 4]) = acc;
        }
        __syncthreads();
        // ---- stage 3: last transposed conv + 4x4 average pooling = 3x3 / stride 2 / pad 1 over a2
        for (int o = tid; o < 289; o += 256) {
            const int y = o / 17, x = o - y * 17;
            float acc = b3;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int iy = 2 * y + u - 1;
                if ((unsigned)iy > 32u) continue;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int ix = 2 * x + v - 1;
                    if ((unsigned)ix > 32u) continue;
                    const float4 *a = reinterpret_cast<const float4 *>(&a2s[(iy * 33 + ix) * 16]);
                    const float4 *k = reinterpret_cast<const float4 *>(&Ks[(u * 3 + v) * 16]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 av = a[c], kv = k[c];
                        acc = fmaf(av.x, kv.x, acc); acc = fmaf(av.y, kv.y, acc);
                        acc = fmaf(av.z, kv.z, acc); acc = fmaf(av.w, kv.w, acc);
                    }
                }
            }
            frames[(size_t)f * 289 + o] = acc;
        }
    }
}

// LSTM cell, pointwise part (torch gate order i, f, g, o): c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c').
// The pre-activations are the sum of up to three terms, added here instead of in GEMM epilogues / copy kernels:
// gates_a float[B][4H] (a GEMM result), gates_b (optional, rows ldb floats apart: the slice of one time step out of
// the input projections of all known steps) and bias (optional, float[4H]).  One pass instead of eight elementwise
// launches plus the strided copy torch.addmm makes of a non-contiguous addend.
__global__ __launch_bounds__(256) void ppo_lstm_cell_kernel(const float4 *__restrict__ ga, const float4 *__restrict__ gb,
                                                            long long ldb4, const float4 *__restrict__ bias,
                                                            float4 *__restrict__ c, float4 *__restrict__ h, int B, int H4) {
    const int i = blockIdx.x * 256 + threadIdx.x;                 // one float4 of one row's hidden vector
    if (i >= B * H4) return;
    const int b = i / H4, j = i - b * H4;
    float4 gi = make_float4(0.f, 0.f, 0.f, 0.f), gf = gi, gg = gi, go = gi;
    if (ga) {
        const float4 *g = ga + (size_t)b * 4 * H4;
        gi = g[j]; gf = g[H4 + j]; gg = g[2 * H4 + j]; go = g[3 * H4 + j];
    }
    auto add4 = [](float4 &a, const float4 &v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    if (gb) {
        const float4 *q = gb + (size_t)b * ldb4;
        add4(gi, q[j]); add4(gf, q[H4 + j]); add4(gg, q[2 * H4 + j]); add4(go, q[3 * H4 + j]);
    }
    if (bias) { add4(gi, bias[j]); add4(gf, bias[H4 + j]); add4(gg, bias[2 * H4 + j]); add4(go, bias[3 * H4 + j]); }
    float4 cv = c[i], hv;
    auto sig = [](float x) { return 1.0f / (1.0f + __expf(-x)); };
#define LSTM_LANE(m)                                                                                                      \
    cv.m = sig(gf.m) * cv.m + sig(gi.m) * tanhf(gg.m);                                                                   \
    hv.m = sig(go.m) * tanhf(cv.m);
    LSTM_LANE(x) LSTM_LANE(y) LSTM_LANE(z) LSTM_LANE(w)
#undef LSTM_LANE
    c[i] = cv;
    h[i] = hv;
}

int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TW_OK : TW_E_HIP;
}

}  // namespace

extern "C" {

int ppo_sample(const float *probs, int B, int A, const float *uniforms, uint64_t seed, uint64_t offset,
               int32_t *action, float *logp, void *stream) {
    return ppo_sample_dev(probs, B, A, uniforms, seed, offset, nullptr, action, logp, stream);
}

int ppo_sample_dev(const float *probs, int B, int A, const float *uniforms, uint64_t seed, uint64_t offset,
                   const uint64_t *offset_dev, int32_t *action, float *logp, void *stream) {
    if (!probs || !action || !logp || B <= 0) return TW_E_ARG;
    const dim3 grid((B + 255) / 256), block(256);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    hipStream_t st = (hipStream_t)stream;
#define PPO_SAMPLE_LAUNCH(AA) \
    hipLaunchKernelGGL(ppo_sample_kernel<AA>, grid, block, 0, st, probs, B, uniforms, k0, k1, offset, offset_dev, action, logp)
    switch (A) {
    case 5: PPO_SAMPLE_LAUNCH(5); break;
    case 2: PPO_SAMPLE_LAUNCH(2); break;
    case 3: PPO_SAMPLE_LAUNCH(3); break;
    case 4: PPO_SAMPLE_LAUNCH(4); break;
    case 7: PPO_SAMPLE_LAUNCH(7); break;
    default: return TW_E_ARG;
    }
#undef PPO_SAMPLE_LAUNCH
    return check_launch();
}

int ppo_gae(const float *reward, const float *value, const float *next_value, const uint8_t *done, float gamma,
            float lambda, int use_done_mask, int T, int N, float *adv, float *target, float *ret, void *stream) {
    if (!reward || !value || !next_value || T <= 0 || N <= 0 || (use_done_mask && !done)) return TW_E_ARG;
    if (N >= 16384)
        hipLaunchKernelGGL(ppo_gae_kernel<64>, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, reward, value,
                           next_value, done, gamma, lambda, use_done_mask, T, N, adv, target, ret);
    else
        hipLaunchKernelGGL(ppo_gae_kernel<16>, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, reward, value,
                           next_value, done, gamma, lambda, use_done_mask, T, N, adv, target, ret);
    return check_launch();
}

int ppo_adv_norm(float *adv, int64_t n, float eps, double *workspace, void *stream) {
    if (!adv || n <= 0 || !workspace) return TW_E_ARG;
    int nblocks = (int)((n + 1023) / 1024);                  // partial sums: one workgroup per CU at most
    if (nblocks > 256) nblocks = 256;
    int nblocks2 = (int)((n + 255) / 256);
    if (nblocks2 > 2048) nblocks2 = 2048;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ppo_moments_kernel, dim3(nblocks), dim3(256), 0, st, adv, n, workspace);
    hipLaunchKernelGGL(ppo_normalise_kernel, dim3(nblocks2), dim3(256), 0, st, adv, n, eps, workspace, nblocks);
    return check_launch();
}

int ppo_loss_fwd_bwd(const float *probs, const int32_t *action, const float *old_logp, const float *adv,
                     const float *value, const float *target_v, int B, int A, float clip, float ent_coef,
                     float *losses, float *grad_probs, float *grad_value, float *workspace, void *stream) {
    return ppo_loss_fwd_bwd_masked(probs, action, old_logp, adv, value, target_v, B, B, A, clip, ent_coef, losses,
                                   grad_probs, grad_value, workspace, stream);
}

int ppo_loss_fwd_bwd_masked(const float *probs, const int32_t *action, const float *old_logp, const float *adv,
                            const float *value, const float *target_v, int B, int n_valid, int A, float clip,
                            float ent_coef, float *losses, float *grad_probs, float *grad_value, float *workspace,
                            void *stream) {
    if (!probs || !action || !old_logp || !adv || !value || !target_v || !losses || !grad_probs || !grad_value ||
        !workspace || B <= 0 || A != 5 || n_valid <= 0 || n_valid > B)
        return TW_E_ARG;
    const int nblocks = (B + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ppo_loss_kernel<5>, dim3(nblocks), dim3(256), 0, st, probs, action, old_logp, adv, value,
                       target_v, B, n_valid, clip, ent_coef, grad_probs, grad_value, workspace);
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, nblocks, n_valid, losses);
    return check_launch();
}

int ppo_gather_stack(const float *frames, int frame_pitch, const float *pos_frames, int N, const int32_t *k_idx,
                     const int32_t *n_idx, const int32_t *age, const float *init_frame, const float *init_pos, int B,
                     float *out, float *pos_out, void *stream) {
    if (!frames || !k_idx || !n_idx || !age || !init_frame || !out || B <= 0 || frame_pitch < TW_CELLS) return TW_E_ARG;
    if (pos_out && (!pos_frames || !init_pos)) return TW_E_ARG;
    hipLaunchKernelGGL(ppo_gather_stack_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, frames, frame_pitch,
                       pos_frames, N, k_idx, n_idx, age, init_frame, init_pos, B, out, pos_out);
    return check_launch();
}

int ppo_gather_stack_u8(const uint8_t *frames, int frame_pitch, const float *pos_frames, int N, const int32_t *k_idx,
                        const int32_t *n_idx, const int32_t *age, const float *init_frame, const float *init_pos, int B,
                        float *out, float *pos_out, void *stream) {
    if (!frames || !k_idx || !n_idx || !age || !init_frame || !out || B <= 0 || frame_pitch < TW_CELLS) return TW_E_ARG;
    if (pos_out && (!pos_frames || !init_pos)) return TW_E_ARG;
    hipLaunchKernelGGL(ppo_gather_stack_kernel<uint8_t>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, frames,
                       frame_pitch, pos_frames, N, k_idx, n_idx, age, init_frame, init_pos, B, out, pos_out);
    return check_launch();
}

int ppo_her_relabel_window(const float *pos, const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0,
                           const float *reward, const int32_t *choices, uint64_t seed, uint32_t env_id0, uint32_t step0,
                           int T, int N, int max_goals, int skip, const int64_t *offsets, int32_t *counts, int32_t *out_t,
                           int32_t *out_n, float *out_goal, float *out_reward, uint8_t *out_done, void *stream) {
    if (!pos || !terminated || !truncated || !age0 || !reward || T <= 0 || N <= 0 || max_goals < 0) return TW_E_ARG;
    if (max_goals > 4) return TW_E_ARG;                                  // 4 pick columns / one Philox call
    if (skip < 0 || skip >= HER_MAX_LEN) return TW_E_ARG;
    if (!offsets && !counts) return TW_E_ARG;
    if (offsets && (!out_t || !out_n || !out_goal || !out_reward || !out_done)) return TW_E_ARG;
    hipLaunchKernelGGL(ppo_her_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, pos, terminated, truncated, age0,
                       reward, choices, (uint32_t)seed, (uint32_t)(seed >> 32), env_id0, step0, T, N, max_goals, skip,
                       offsets, counts, out_t, out_n, out_goal, out_reward, out_done);
    return check_launch();
}

int ppo_her_relabel(const float *pos, const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0,
                    const float *reward, const int32_t *choices, uint64_t seed, uint32_t env_id0, uint32_t step0, int T,
                    int N, int max_goals, const int64_t *offsets, int32_t *counts, int32_t *out_t, int32_t *out_n,
                    float *out_goal, float *out_reward, uint8_t *out_done, void *stream) {
    return ppo_her_relabel_window(pos, terminated, truncated, age0, reward, choices, seed, env_id0, step0, T, N, max_goals,
                                  0, offsets, counts, out_t, out_n, out_goal, out_reward, out_done, stream);
}

int ppo_age_scan(const uint8_t *terminated, const uint8_t *truncated, const int32_t *age0, int T, int N, int32_t *age,
                 void *stream) {
    if (!terminated || !truncated || !age0 || !age || T <= 0 || N <= 0) return TW_E_ARG;
    hipLaunchKernelGGL(ppo_age_scan_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, terminated,
                       truncated, age0, T, N, age);
    return check_launch();
}

int ppo_bias_relu_nhwc(float *y, const float *bias, int64_t n_pixels, int C, void *stream) {
    if (!y || !bias || n_pixels <= 0 || C <= 0 || (C & 3) || ((uintptr_t)y & 15u) || ((uintptr_t)bias & 15u)) return TW_E_ARG;
    const size_t n4 = (size_t)n_pixels * (C / 4);
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(ppo_bias_relu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4 *>(y),
                       reinterpret_cast<const float4 *>(bias), n4, C / 4);
    return check_launch();
}

int ppo_relu_bwd_bias_grad_nhwc_blocks(int64_t n_pixels, int C) {
    if (n_pixels <= 0 || C <= 0 || (C & 3) || C > 256) return TW_E_ARG;
    const int64_t want = (n_pixels + 511) / 512;              // >= 512 pixels per block
    return (int)(want < 4096 ? want : 4096);
}

int ppo_relu_bwd_bias_grad_nhwc(const float *gy, const float *y, float *gx, float *partial, int64_t n_pixels, int C,
                                void *stream) {
    const int blocks = ppo_relu_bwd_bias_grad_nhwc_blocks(n_pixels, C);
    if (blocks <= 0 || !gy || !y || !gx || !partial || (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gx | (uintptr_t)partial) & 15u))
        return TW_E_ARG;
    const int ppb = (int)((n_pixels + blocks - 1) / blocks);
    hipLaunchKernelGGL(ppo_relu_bwd_bias_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(gy), reinterpret_cast<const float4 *>(y),
                       reinterpret_cast<float4 *>(gx), reinterpret_cast<float4 *>(partial), (size_t)n_pixels, C / 4, ppb);
    return check_launch();
}

int ppo_conv1_up4_bias_relu(const float *frames, int B, int F, const float *folded_w, const float *bias, float *out,
                            void *stream) {
    return ppo_conv1_up4_bias_relu_c(frames, B, F, 64, folded_w, bias, out, stream);
}

int ppo_conv1_up4_bias_relu_c(const float *frames, int B, int F, int C_out, const float *folded_w, const float *bias,
                              float *out, void *stream) {
    if (!frames || !folded_w || !bias || !out || B <= 0 || ((uintptr_t)out & 15u) || ((uintptr_t)folded_w & 15u) ||
        ((uintptr_t)bias & 15u))
        return TW_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define PPO_CONV1_LAUNCH(FF, CQQ)                                                                                        \
    hipLaunchKernelGGL((ppo_conv1_up4_kernel<FF, CQQ>), dim3(B), dim3(256), 0, st, frames,                               \
                       reinterpret_cast<const float4 *>(folded_w), reinterpret_cast<const float4 *>(bias),               \
                       reinterpret_cast<float4 *>(out), B)
    if (F == 4 && C_out == 64) PPO_CONV1_LAUNCH(4, 16);
    else if (F == 8 && C_out == 64) PPO_CONV1_LAUNCH(8, 16);
    else if (F == 1 && C_out == 16) PPO_CONV1_LAUNCH(1, 4);
    else return TW_E_ARG;
#undef PPO_CONV1_LAUNCH
    return check_launch();
}

int ppo_conv1_up4_bwd_groups(int B) { return B <= 0 ? TW_E_ARG : (B < 256 ? B : 256); }

int ppo_conv1_up4_bwd(const float *frames, int B, int F, const float *gy, const float *y, float *gw_partial,
                      float *gb_partial, void *stream) {
    const int groups = ppo_conv1_up4_bwd_groups(B);
    if (groups <= 0 || !frames || !gy || !y || !gw_partial || !gb_partial ||
        (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)gw_partial | (uintptr_t)gb_partial) & 15u))
        return TW_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(groups, 4), block(256);
    if (F == 4)
        hipLaunchKernelGGL(ppo_conv1_up4_bwd_kernel<4>, grid, block, 0, st, frames, reinterpret_cast<const float4 *>(gy),
                           reinterpret_cast<const float4 *>(y), reinterpret_cast<float4 *>(gw_partial),
                           reinterpret_cast<float4 *>(gb_partial), B);
    else if (F == 8)
        hipLaunchKernelGGL(ppo_conv1_up4_bwd_kernel<8>, grid, block, 0, st, frames, reinterpret_cast<const float4 *>(gy),
                           reinterpret_cast<const float4 *>(y), reinterpret_cast<float4 *>(gw_partial),
                           reinterpret_cast<float4 *>(gb_partial), B);
    else
        return TW_E_ARG;
    return check_launch();
}

int ppo_decoder_frames(const float *z, int n_frames, const float *w1, const float *b1, const float *w2, const float *b2,
                       const float *kfold, float b3, float *frames, void *stream) {
    if (!z || !w1 || !b1 || !w2 || !b2 || !kfold || !frames || n_frames <= 0) return TW_E_ARG;
    const int grid = n_frames < 512 ? n_frames : 512;          // 120 KB of LDS: one workgroup per CU, two rounds of them
    hipLaunchKernelGGL(ppo_decoder_frames_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, n_frames, w1, b1, w2,
                       b2, kfold, b3, frames);
    return check_launch();
}

int ppo_lstm_cell(const float *gates_a, const float *gates_b, long long ldb, const float *bias, float *c, float *h, int B,
                  int H, void *stream) {
    if ((!gates_a && !gates_b) || !c || !h || B <= 0 || H <= 0 || (H & 3) ||
        (((uintptr_t)gates_a | (uintptr_t)c | (uintptr_t)h) & 15u))
        return TW_E_ARG;
    if (gates_b && ((((uintptr_t)gates_b) & 15u) || (ldb & 3) || ldb < 4LL * H)) return TW_E_ARG;
    if (bias && (((uintptr_t)bias) & 15u)) return TW_E_ARG;
    const long long n = (long long)B * (H / 4);
    if (n > 0x7fffffffLL) return TW_E_ARG;
    hipLaunchKernelGGL(ppo_lstm_cell_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(gates_a), reinterpret_cast<const float4 *>(gates_b), ldb / 4,
                       reinterpret_cast<const float4 *>(bias), reinterpret_cast<float4 *>(c), reinterpret_cast<float4 *>(h),
                       B, H / 4);
    return check_launch();
}

}  // extern "C"
