// twoarmy_engine.hip -- MI355X (gfx950) MiniGrid-Twoarmy step / observation engine.
//
// Design.  The MI355X scalar unit issues one instruction per cycle per CU,
// the four SIMDs together two wave64 VALU instructions per cycle, so wave-uniform "scalar" env
// logic is the slowest possible formulation.  The kernel therefore splits every env step into
//
//   LOGIC   lane-per-env: one wavefront (= one 64-thread workgroup) owns E consecutive envs and
//           lanes 0..E-1 run the whole transition (ball / patrol moves, agent move, wall drop,
//           patrol spawn, shaped reward, risk counter, episode-end re-arm, Philox draws) on VGPRs;
//   EMIT    wave-per-env: all 64 lanes cooperate on one env at a time to produce its outputs:
//             * the uint8[V][V][3] egocentric image as 16-byte chunks packed in registers from a
//               wall-padded 33x33 LDS copy of the grid (6 ds_read_b32 + 11 VALU per lane, one
//               global_store_dwordx4 per lane, no bounds checks, no LDS byte staging),
//             * the fp32 289-cell state matrix straight from an LDS-resident float image
//               (ds_read_b128 + agent-cell patch + global_store_dwordx4),
//           with a generic byte-staging path for unaligned/dense destinations and agent_dir != 3.
//
// Grid planes live in HBM as SoA uint8[N][289] type / colour planes plus one int32[48] record per
// env; they are staged into LDS once per launch and stay there for all T steps of a rollout.
// Done envs are found with a wavefront ballot and re-generated in place (auto-reset).
//
// Behavioural contract (bit-exact vs the reference, checked against oracle/ + tests/golden):
//   gym_minigrid/envs/twoarmy_v6.py:83-325, twoarmy_v4.py:82-322  (Twoarmy step)
//   gym_minigrid/minigrid.py:1333-1441 (MiniGridEnv.step), :1262-1293, :1443-1496, :641-660,
//   :627-639, :749-772 (view extents, slice, rotate_left, encode)
//   soa/env_buffer.py:300-334 (matrix_env, data_env)
// From-scratch closed-form implementation; shares no code with oracle/.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "twoarmy.h"

namespace {

constexpr int GS = TW_GRID;
constexpr int NC = TW_CELLS;
constexpr int REC = TW_REC_WORDS;

// packed cell code: type | colour << 8 | state << 16   (OBJECT_TO_IDX / COLOR_TO_IDX, minigrid.py:40-67)
constexpr uint32_t C_EMPTY = 1u;
constexpr uint32_t C_WALL = 2u | (5u << 8);
constexpr uint32_t C_BALL = 6u | (4u << 8);
constexpr uint32_t C_GOAL = 8u | (1u << 8);

// state-matrix values (soa/env_buffer.py:305-316) as fp32 bit patterns
constexpr uint32_t M_FREE = 0x3f666666u;   //  0.9f
constexpr uint32_t M_WALL = 0xbf666666u;   // -0.9f
constexpr uint32_t M_BALL = 0xbf000000u;   // -0.5f
constexpr uint32_t M_AGENT = 0x3e99999au;  //  0.3f

// per-env LDS image: wall-padded grid codes, x in -8..24, y in -16..16, stored x-major ("transposed")
// with a y pitch of 35 words, plus the float matrix image.  The observation walks view cells in
// (x, y) order, so x-major makes a lane's six cells (nearly) consecutive words; pitch 35 makes the six
// gather instructions at most 2-way bank conflicted (row-major pitch 33 was 11-way: SQ_LDS_BANK_CONFLICT
// was 41 % of the kernel's cycles).
constexpr int GPW = 33, GPX0 = 8, GPY0 = 16, GPP = 35;
constexpr int GP_WORDS = GPW * GPP;        // 1155
constexpr int MAT_OFF = 1156;              // 16-byte aligned
constexpr int MAT_WORDS = 292;             // 289 + 3 pad (zeros)
constexpr int MATC_OFF = MAT_OFF + MAT_WORDS;    // 1448: byte image of the matrix as 2-bit codes (TW_F_MATRIX_CODE)
constexpr int MATC_BYTES = 304;                 // 289 + pad to 16
constexpr int ENV_WORDS = MATC_OFF + MATC_BYTES / 4;   // 1524
constexpr int STAGE_WORDS = 220;
// record layout of the outputs (tw_alloc_outputs): one block per env-step = 292 floats of matrix, then the image row
// (V*V*3 bytes rounded up to 16): 1168 + 880 = 2048 bytes for V = 17
constexpr int REC_OBS_OFF = MAT_WORDS * 4;
static_assert(REC_OBS_OFF + 880 == 2048, "matrix row + view-17 image row are exactly 2048 bytes");

enum { R_STEP = 0, R_RISK = 1, R_HIT = 2, R_ROOM2 = 3, R_GOAL = 4 };

struct Params {
    uint8_t *type;            // state read from here ...
    uint8_t *colour;
    int32_t *rec;
    uint8_t *type_out;        // ... and written here (== the inputs for an in-place launch)
    uint8_t *colour_out;
    int32_t *rec_out;
    int *abnormal;            // device flag raised by the pipelined kernel when an env leaves normal play
    int *abnormal_other;      // the flag of the previous pipelined launch: cleared here instead of a memset node
    int only_if_flagged;      // sequential kernel: run only if *abnormal != 0 (fallback launch)
    int *fb_count;            // engine statistic: pipelined launches that had to be re-run by the fallback launch
    int n_envs;
    int view;
    int variant;
    uint32_t seed_lo, seed_hi;
    uint32_t env_id0;
    int T;
    const int32_t *actions;   // [T][N] or null
    const uint32_t *draws;    // [T][N][8] or null
    uint8_t *obs;
    float *matrix;
    float *pos;
    float *reward;
    uint8_t *term;
    uint8_t *trunc;
    int obs_pitch;            // bytes between consecutive envs' images
    int mat_pitch;            // floats between consecutive envs' matrices
    int flags;
    int record;               // record layout: one 2048-byte block per env-step = 1168 B matrix row | 880 B image row
    const uint32_t *pipe_tab; // per-engine constant tables of the pipelined kernel (tw_pipe_tables_kernel)
};

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t &c0, uint32_t &c1,
                                              uint32_t &c2, uint32_t &c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const uint32_t lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// all four words of one draw block: counter = (env_id, t, block, 'TWOA')
__device__ __forceinline__ void draw_block(uint32_t k0, uint32_t k1, uint32_t env_id, uint32_t t, uint32_t block,
                                           uint32_t (&w)[4]) {
    w[0] = env_id; w[1] = t; w[2] = block; w[3] = 0x54574F41u;
    philox4x32_10(k0, k1, w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------- grid generation
// Twoarmy_v{4,6}._gen_grid (twoarmy_v6.py:39-81): closed form of the initial cell at (x, y).
__device__ __forceinline__ uint32_t gen_cell(int x, int y) {
    if (x == 0 || y == 0 || x == GS - 1 || y == GS - 1) return C_WALL;
    if (y == 8) {
        if (x <= 5 || x >= 11) return C_WALL;
        if (x >= 7 && x <= 9) return C_BALL;
        return C_EMPTY;
    }
    if (x == 14 && y == 2) return C_GOAL;
    return C_EMPTY;
}

__device__ __forceinline__ uint32_t mat_of_code(uint32_t code) {   // env_buffer.py:306-314
    const uint32_t t = code & 0xffu;
    return t == 2u ? M_WALL : (t == 6u ? M_BALL : M_FREE);
}

// per-lane env state (names follow the reference attributes)
struct EnvS {
    int ax, ay, dir, step_count, step_move, m6, m4, pone, patrol, up1, right2, upd_long, upd_horiz, risk,
        first_room2;
    int obx[3], oby[3], o1x[3], o1y[3], o1v, o2x[4], o2y[4], o2v, gx, gy;
    uint32_t t;
    int err, max_steps, episodes, last_reward, last_term, last_trunc;
    int wall_i1, wall_i2;     // offsets of the two dropped 2x2 wall blocks (valid while pone)
};

__device__ __forceinline__ void load_env(EnvS &s, const int32_t *r) {
    s.ax = r[TW_AX]; s.ay = r[TW_AY]; s.dir = r[TW_DIR];
    s.step_count = r[TW_STEP_COUNT]; s.step_move = r[TW_STEP_MOVE];
    s.m6 = (int)((uint32_t)s.step_move % 6u); s.m4 = s.step_move & 3;
    s.pone = r[TW_PONE]; s.patrol = r[TW_PATROL]; s.up1 = r[TW_UP1];
    s.right2 = r[TW_RIGHT2]; s.upd_long = r[TW_UPD_LONG]; s.upd_horiz = r[TW_UPD_HORIZ];
    s.risk = r[TW_RISK]; s.first_room2 = r[TW_FIRST_ROOM2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.obx[k] = r[TW_OBX + k]; s.oby[k] = r[TW_OBY + k];
        s.o1x[k] = r[TW_O1X + k]; s.o1y[k] = r[TW_O1Y + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.o2x[k] = r[TW_O2X + k]; s.o2y[k] = r[TW_O2Y + k]; }
    s.o1v = r[TW_O1_VALID]; s.o2v = r[TW_O2_VALID];
    s.gx = r[TW_GOAL_X]; s.gy = r[TW_GOAL_Y];
    s.t = (uint32_t)r[TW_T]; s.err = r[TW_ERROR]; s.max_steps = r[TW_MAX_STEPS];
    s.episodes = r[TW_EPISODES]; s.last_reward = r[TW_LAST_REWARD];
    s.last_term = r[TW_LAST_TERM]; s.last_trunc = r[TW_LAST_TRUNC];
    s.wall_i1 = r[TW_WALL_I1]; s.wall_i2 = r[TW_WALL_I2];
}

__device__ __forceinline__ void store_env(const EnvS &s, int32_t *r) {
    r[TW_AX] = s.ax; r[TW_AY] = s.ay; r[TW_DIR] = s.dir;
    r[TW_STEP_COUNT] = s.step_count; r[TW_STEP_MOVE] = s.step_move;
    r[TW_PONE] = s.pone; r[TW_PATROL] = s.patrol; r[TW_UP1] = s.up1; r[TW_RIGHT2] = s.right2;
    r[TW_UPD_LONG] = s.upd_long; r[TW_UPD_HORIZ] = s.upd_horiz; r[TW_RISK] = s.risk;
    r[TW_FIRST_ROOM2] = s.first_room2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r[TW_OBX + k] = s.obx[k]; r[TW_OBY + k] = s.oby[k];
        r[TW_O1X + k] = s.o1x[k]; r[TW_O1Y + k] = s.o1y[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { r[TW_O2X + k] = s.o2x[k]; r[TW_O2Y + k] = s.o2y[k]; }
    r[TW_O1_VALID] = s.o1v; r[TW_O2_VALID] = s.o2v;
    r[TW_GOAL_X] = s.gx; r[TW_GOAL_Y] = s.gy;
    r[TW_T] = (int32_t)s.t; r[TW_ERROR] = s.err; r[TW_MAX_STEPS] = s.max_steps;
    r[TW_EPISODES] = s.episodes; r[TW_LAST_REWARD] = s.last_reward;
    r[TW_LAST_TERM] = s.last_term; r[TW_LAST_TRUNC] = s.last_trunc;
    r[TW_WALL_I1] = s.wall_i1; r[TW_WALL_I2] = s.wall_i2;
}

// MiniGridEnv.reset + _gen_grid on the scalar side (minigrid.py:947-980, twoarmy_v6.py:56-77).
__device__ __forceinline__ void reset_scalars(EnvS &s) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.obx[k] = 7 + k; s.oby[k] = 8; }
    s.o1v = 0; s.o2v = 0;
    s.ax = 3; s.ay = 15; s.dir = 3; s.gx = 14; s.gy = 2;
    s.step_count = 0; s.err = 0;
}

// Invariants of the kernel's TURBO step (see tw_rollout_kernel): the three row-8 balls are an adjacent
// triple well inside the grid, the agent faces up and stands strictly inside the border.
__device__ __forceinline__ bool normal_mode(const EnvS &s) {
    return (s.oby[0] == 8) & (s.oby[1] == 8) & (s.oby[2] == 8) & (s.obx[1] == s.obx[0] + 1) &
           (s.obx[2] == s.obx[0] + 2) & ((unsigned)(s.obx[0] - 2) <= 10u) & (s.dir == 3) &
           ((unsigned)(s.ax - 1) <= 14u) & ((unsigned)(s.ay - 1) <= 14u);
}

__device__ __forceinline__ bool inb(int x, int y) { return (unsigned)x < (unsigned)GS && (unsigned)y < (unsigned)GS; }
__device__ __forceinline__ int gpi(int x, int y) { return (x + GPX0) * GPP + y + GPY0; }

// One workgroup == one wavefront: LDS operations of a wave execute in issue order, so cross-lane
// LDS hand-offs only need the COMPILER to keep program order.  __syncthreads() would also emit
// s_waitcnt vmcnt(0) and drain every outstanding global store (microseconds per step under load).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16-byte global store of the emission path.  Plain stores on purpose: non-temporal ones (streaming hint) were
// measured 30 % slower (0.318 vs 0.242 ms per launch on one box) -- the L2 write-combining matters here.
__device__ __forceinline__ void store16(void *dst, const uint4 &v) { *reinterpret_cast<uint4 *>(dst) = v; }

// matrix value -> code of the reduced-precision frame plane: 0 free/goal (0.9), 1 wall (-0.9), 2 ball (-0.5), 3 agent (0.3)
__device__ __forceinline__ uint8_t mcode_of(uint32_t mval) {
    return mval == M_WALL ? 1 : (mval == M_BALL ? 2 : (mval == M_AGENT ? 3 : 0));
}
__device__ __forceinline__ uint8_t *mcb(uint32_t *env) { return reinterpret_cast<uint8_t *>(env + MATC_OFF); }
__device__ __forceinline__ const uint8_t *mcb(const uint32_t *env) { return reinterpret_cast<const uint8_t *>(env + MATC_OFF); }

// Grid.set on the env's LDS image: code plane + float matrix image + matrix code bytes (no bounds check here).
__device__ __forceinline__ void put(uint32_t *env, int x, int y, uint32_t code, uint32_t mval) {
    env[gpi(x, y)] = code;
    env[MAT_OFF + y * GS + x] = mval;
    mcb(env)[y * GS + x] = mcode_of(mval);
}

// Patrol group move: clear every cell, then put each ball at +d inside try/except (twoarmy_v4.py:119-176).
template <int NB>
__device__ __forceinline__ bool move_group(uint32_t *env, int (&xs)[NB], int (&ys)[NB], int valid, int dx, int dy,
                                           int &err) {
    if (!valid) { err = TW_ENV_TYPE; return false; }      // cur_pos is None
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (ok) { if (inb(xs[k], ys[k])) put(env, xs[k], ys[k], C_EMPTY, M_FREE); else ok = false; }
    if (!ok) { err = TW_ENV_ASSERT; return false; }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int nx = xs[k] + dx, ny = ys[k] + dy;
        if (inb(nx, ny)) { put(env, nx, ny, C_BALL, M_BALL); xs[k] = nx; ys[k] = ny; }
    }
    return true;
}

__device__ __forceinline__ float reward_value(int code) {
    return code == R_STEP ? -0.01f : code == R_RISK ? -0.1f : code == R_HIT ? -0.9f : code == R_ROOM2 ? 0.2f : 0.9f;
}

// ---------------------------------------------------------------- observation: generic path
// gen_obs_grid + encode (minigrid.py:1443-1496) in closed form.
//   view cell (i, j) (i = x index of the rotated view, j = y index; output byte (i*V + j)*3 + ch)
//   maps to slice coords   dir 0: (V-1-j, i)   1: (V-1-i, V-1-j)   2: (j, V-1-i)   3: (i, j)
//   and slice origin       dir 0: (ax, ay-h)   1: (ax-h, ay)       2: (ax-V+1, ay-h)   3: (ax-h, ay-V+1)
// Out-of-grid cells read as Wall (2,5,0) (minigrid.py:655-656); the agent's own cell (V/2, V-1) is
// forced empty (minigrid.py:1472-1476, carrying is always None in Twoarmy).
// The image is staged in LDS at the same 4-byte phase as its global destination and copied out as
// aligned dwords (+ <= 3 head / tail bytes).  Works for any destination alignment and any dir.
template <typename Fetch>
__device__ __forceinline__ void emit_obs_generic(Fetch fetch, uint32_t *stage, int lane, int V, int ax, int ay,
                                                 int dir, uint8_t *dst) {
    const int VV = V * V, h = V >> 1, nb = VV * 3;
    int topx, topy;
    if (dir == 0) { topx = ax; topy = ay - h; }
    else if (dir == 1) { topx = ax - h; topy = ay; }
    else if (dir == 2) { topx = ax - V + 1; topy = ay - h; }
    else { topx = ax - h; topy = ay - V + 1; }
    const int c_agent = h * V + V - 1;
    const uint32_t off = (uint32_t)(uintptr_t)dst & 3u;
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage) + off;
    for (int c = lane; c < VV; c += 64) {
        const int i = c / V, j = c - i * V;
        int sx, sy;
        if (dir == 0) { sx = V - 1 - j; sy = i; }
        else if (dir == 1) { sx = V - 1 - i; sy = V - 1 - j; }
        else if (dir == 2) { sx = j; sy = V - 1 - i; }
        else { sx = i; sy = j; }
        const int x = topx + sx, y = topy + sy;
        uint32_t v = inb(x, y) ? fetch(x, y) : C_WALL;
        if (c == c_agent) v = C_EMPTY;
        sb[3 * c + 0] = (uint8_t)v;
        sb[3 * c + 1] = (uint8_t)(v >> 8);
        sb[3 * c + 2] = (uint8_t)(v >> 16);
    }
    wave_sync();
    const int first = (4 - (int)off) & 3;
    const int nmid = (nb - first) >> 2;
    const int tail = (nb - first) & 3;
    if (lane < first) dst[lane] = sb[lane];
    {
        uint32_t *gd = reinterpret_cast<uint32_t *>(dst + first);
        const uint32_t *sd = stage + ((off + first) >> 2);
        for (int d = lane; d < nmid; d += 64) gd[d] = sd[d];
    }
    if (lane < tail) dst[first + nmid * 4 + lane] = sb[first + nmid * 4 + lane];
    wave_sync();
}

// ---------------------------------------------------------------- observation: fast path
// agent_dir == 3 (always, in Twoarmy) and a 16-byte aligned destination.  Lane l owns output
// bytes [16l, 16l+16): they cover view cells c0 .. c0+5 with c0 = floor(16l/3); view cell
// c = i*V + j lies at padded-grid offset (j-(V-1))*33 + (i-h) from the agent's own cell, so the
// six gathers are one ds_read_b32 each at (agent cell address + per-lane constant).  The 24-bit
// codes are packed into a 144-bit stream and funnel-shifted by the lane's byte phase (16l mod 3).
// and/or masks patch the agent's own view cell to (1,0,0) and zero the bytes past V*V*3.
struct ObsFast {
    int rel4[6];
    uint32_t shift;
    uint32_t andm[4], orm[4];
    int active;
    int chunk;                // 16-byte chunk of the image this lane produces (== lane, or lane - 9 in the record layout)
};

// chunk = index of the 16-byte piece of the image the calling lane owns (negative or past the image: idle lane)
__device__ __forceinline__ ObsFast make_obs_fast(int chunk, int V) {
    ObsFast f;
    const int VV = V * V, h = V >> 1, nb = VV * 3;
    const int nchunks = (nb + 15) >> 4;
    f.active = chunk >= 0 && chunk < nchunks;
    f.chunk = f.active ? chunk : 0;
    const int lane = f.chunk;
    const int s0 = 16 * lane;
    const int c0 = s0 / 3;
    f.shift = 8u * (uint32_t)(s0 - 3 * c0);
    const int c_agent = h * V + V - 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int c = c0 + k;
        if (c >= VV) c = 0;
        const int i = c / V, j = c - i * V;
        f.rel4[k] = f.active ? 4 * ((i - h) * GPP + (j - (V - 1))) : 0;   // idle lanes read the agent cell
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t am = 0, om = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sbyte = s0 + 4 * w + b;
            const int c = sbyte / 3, ch = sbyte - 3 * c;
            if (sbyte >= nb) continue;                              // zero the pad bytes
            if (c == c_agent) { if (ch == 0) om |= 1u << (8 * b); } // (1,0,0)
            else am |= 0xffu << (8 * b);
        }
        f.andm[w] = am; f.orm[w] = om;
    }
    return f;
}

__device__ __forceinline__ void emit_obs_fast(const uint32_t *env, int ax, int ay, const ObsFast &f, int lane,
                                              uint8_t *dst) {
    if (!f.active) return;
    const char *base = reinterpret_cast<const char *>(env + gpi(ax, ay));
    uint32_t c[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) c[k] = *reinterpret_cast<const uint32_t *>(base + f.rel4[k]);
    const uint32_t w0 = c[0] | (c[1] << 24);
    const uint32_t w1 = (c[1] >> 8) | (c[2] << 16);
    const uint32_t w2 = (c[2] >> 16) | (c[3] << 8);
    const uint32_t w3 = c[4] | (c[5] << 24);
    const uint32_t w4 = c[5] >> 8;
    uint4 o;
    o.x = (__builtin_amdgcn_alignbit(w1, w0, f.shift) & f.andm[0]) | f.orm[0];
    o.y = (__builtin_amdgcn_alignbit(w2, w1, f.shift) & f.andm[1]) | f.orm[1];
    o.z = (__builtin_amdgcn_alignbit(w3, w2, f.shift) & f.andm[2]) | f.orm[2];
    o.w = (__builtin_amdgcn_alignbit(w4, w3, f.shift) & f.andm[3]) | f.orm[3];
    *reinterpret_cast<uint4 *>(dst + 16 * f.chunk) = o;
}

// ---------------------------------------------------------------- state matrix (env_buffer.py:300-318)
__device__ __forceinline__ void emit_matrix_fast(const uint32_t *env, int ax, int ay, int lane, float *dst) {
    const int ca = ay * GS + ax;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int q = lane + 64 * pass;
        if (q < MAT_WORDS / 4) {
            uint4 v = *reinterpret_cast<const uint4 *>(env + MAT_OFF + 4 * q);
            const int d = ca - 4 * q;
            v.x = d == 0 ? M_AGENT : v.x;
            v.y = d == 1 ? M_AGENT : v.y;
            v.z = d == 2 ? M_AGENT : v.z;
            v.w = d == 3 ? M_AGENT : v.w;
            *reinterpret_cast<uint4 *>(dst + 4 * q) = v;
        }
    }
}

__device__ __forceinline__ void emit_matrix_generic(const uint32_t *env, int ax, int ay, int lane, float *dst) {
    const int ca = ay * GS + ax;
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    for (int c = lane; c < NC; c += 64) d[c] = c == ca ? M_AGENT : env[MAT_OFF + c];
}

// Reduced-precision frame plane (TW_F_MATRIX_CODE): uint8[289] codes instead of float[289]; 4x fewer bytes and exact
// (the four matrix values are a 4-entry LUT).  Fast path: 16-byte aligned destination with a pitch >= 304.
__device__ __forceinline__ void emit_codes(uint32_t *env, int ax, int ay, int lane, uint8_t *dst, bool fast) {
    uint8_t *b = mcb(env);
    const int ca = ay * GS + ax;
    uint8_t old = 0;
    wave_sync();
    if (lane == 0) { old = b[ca]; b[ca] = 3; }                   // the agent's cell
    wave_sync();
    if (fast) {
        if (lane < MATC_BYTES / 16)
            *reinterpret_cast<uint4 *>(dst + 16 * lane) = *reinterpret_cast<const uint4 *>(b + 16 * lane);
    } else {
        for (int c = lane; c < NC; c += 64) dst[c] = b[c];
    }
    wave_sync();
    if (lane == 0) b[ca] = old;
    wave_sync();
}

// (re)generate one env's LDS image interior cooperatively (auto-reset / init)
__device__ __forceinline__ void regen_env(uint32_t *env, int lane) {
    for (int c = lane; c < NC; c += 64) {
        const int y = c / GS, x = c - y * GS;
        const uint32_t code = gen_cell(x, y);
        env[gpi(x, y)] = code;
        env[MAT_OFF + c] = mat_of_code(code);
        mcb(env)[c] = mcode_of(mat_of_code(code));
    }
}

#ifdef TW_STAMP
// Diagnostic build only (make stamp): phase cycle shares via s_memtime; never in the shipped library.
__device__ unsigned long long g_stamp[64][8];
__device__ unsigned long long g_stamp3[64][16][4];   // per wave, absolute s_memtime: kernel entry, loads issued, after the first barrier, first task drawn
__device__ unsigned long long g_stamp2[64][16][3];   // per wave: tasks, poll cycles, work cycles
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
    st_acc[i] += _t - st_prev; st_prev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// ---------------------------------------------------------------- the rollout kernel
// FAST = every output present, native 16-byte layouts, actions supplied, Philox draws: all the
// wave-uniform "is this pointer null / is this layout aligned" branches fold away at compile time.
// Launch bounds: 4 waves per SIMD for E <= 2 (<= 128 VGPRs): at 4096 envs the grid is only 4096 / E
// single-wave workgroups for 1024 SIMDs, and a lone wave issues a dependent instruction every
// ~11 cycles, so residency (not instruction count) sets the step time.
// The whole T-step rollout of E consecutive envs (n0 ..) by ONE wavefront; `lds_env` / `stage` / `recs` / `act_lds` are
// that wavefront's LDS areas.
template <int E, int VARIANT, bool FAST>
__device__ __forceinline__ void rollout_body(const Params &p, const int n0, const int lane, uint32_t *lds_env,
                                             uint32_t *stage, int32_t *recs, int32_t *act_lds) {
    constexpr bool V4 = VARIANT == 4;
    const int N = p.n_envs;
    const int n_mine = n0 + lane;                       // env of this lane (logic phase)
    const bool active = lane < E && n_mine < N;
    const uint32_t env_id = p.env_id0 + (uint32_t)n_mine;
    uint32_t *my_env = lds_env + (lane < E ? lane : 0) * ENV_WORDS;

    // ---- stage records and planes into LDS
    for (int i = lane; i < E * REC; i += 64) {
        const int e = i / REC;
        recs[i] = (n0 + e < N) ? p.rec[(size_t)n0 * REC + i] : 0;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t *env = lds_env + e * ENV_WORDS;
        for (int i = lane; i < GP_WORDS; i += 64) env[i] = C_WALL;          // wall padding
        if (lane < MAT_WORDS - NC) env[MAT_OFF + NC + lane] = 0u;
        if (lane < MATC_BYTES / 4) env[MATC_OFF + lane] = 0u;
        if (lane + 64 < MATC_BYTES / 4) env[MATC_OFF + 64 + lane] = 0u;
    }
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (n0 + e >= N) break;
        uint32_t *env = lds_env + e * ENV_WORDS;
        const size_t gb = (size_t)(n0 + e) * NC;
        for (int c = lane; c < NC; c += 64) {
            const uint32_t code = (uint32_t)p.type[gb + c] | ((uint32_t)p.colour[gb + c] << 8);
            const int y = c / GS, x = c - y * GS;
            env[gpi(x, y)] = code;
            env[MAT_OFF + c] = mat_of_code(code);
            mcb(env)[c] = mcode_of(mat_of_code(code));
        }
    }
    wave_sync();
    EnvS s;
    load_env(s, recs + (lane < E ? lane : 0) * REC);
    bool mode_ok = normal_mode(s);

    const int V = p.view;
    const int obs_chunk_bytes = ((V * V * 3 + 15) >> 4) << 4;
    const bool autoreset = (p.flags & TW_F_AUTORESET) != 0;
    const bool has_obs = FAST || p.obs != nullptr, has_mat = FAST || p.matrix != nullptr;
    const bool has_actions = FAST || p.actions != nullptr, has_draws = !FAST && p.draws != nullptr;
    const bool policy_idx = (p.flags & TW_F_POLICY_IDX) != 0 || !has_actions;
    const bool fast_obs_ok = FAST || (p.obs && ((uintptr_t)p.obs & 15u) == 0 && (p.obs_pitch & 15) == 0 &&
                                      p.obs_pitch >= obs_chunk_bytes);
    const bool fast_mat_ok = FAST || (p.matrix && ((uintptr_t)p.matrix & 15u) == 0 && (p.mat_pitch & 3) == 0 &&
                                      p.mat_pitch >= MAT_WORDS);
    const ObsFast of = make_obs_fast(lane, V);
    const bool code_mode = !FAST && (p.flags & TW_F_MATRIX_CODE) != 0;     // matrix output = uint8 codes, pitch in bytes
    const bool fast_code_ok = code_mode && p.matrix && ((uintptr_t)p.matrix & 15u) == 0 && (p.mat_pitch & 15) == 0 &&
                              p.mat_pitch >= MATC_BYTES;
    const bool layouts_fast = FAST || ((!p.obs || fast_obs_ok) && (!p.matrix || fast_mat_ok || code_mode));
    const unsigned long long m_active = __ballot(active);

    // running output cursors (advance by one time row per step)
    uint8_t *obs_row = has_obs ? p.obs + (size_t)n0 * p.obs_pitch : nullptr;
    float *mat_row = has_mat ? p.matrix + (size_t)n0 * p.mat_pitch : nullptr;
    uint8_t *matc_row = has_mat ? reinterpret_cast<uint8_t *>(p.matrix) + (size_t)n0 * p.mat_pitch : nullptr;
    const size_t obs_step = (size_t)N * p.obs_pitch, mat_step = (size_t)N * p.mat_pitch;
    size_t idx = (size_t)n0 + lane;                         // [t][n] row of this lane's env

#ifdef TW_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory");
#endif
    for (int tt = 0; tt < p.T; ++tt, idx += N, obs_row += obs_step, mat_row += mat_step, matc_row += mat_step) {
        STAMP(0);

        // ---- actions: one coalesced-per-env vector load every 64 steps, parked in LDS
        if (has_actions && (tt & 63) == 0) {
            wave_sync();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int ts = tt + lane;
                act_lds[lane * E + e] = (ts < p.T && n0 + e < N) ? p.actions[(size_t)ts * N + n0 + e] : 0;
            }
            wave_sync();
        }

        STAMP(1);
        // ================= LOGIC part 1 (lane-per-env): everything before gen_obs()
        int err = TW_ENV_OK, have_obs = 0, terminated = 0, truncated = 0, reward = R_STEP;
        uint32_t dw[4] = {0, 0, 0, 0};                      // draw block 0: gate, wall1, wall2, spawn
        const uint32_t t_now = s.t;
        int action = 0;
        if (active) {
            if (has_actions) action = act_lds[(tt & 63) * E + lane];
            else {
                uint32_t w[4];
                draw_block(p.seed_lo, p.seed_hi, env_id, t_now, 2, w);
                action = (int)(w[0] % 5u);
            }
            if (policy_idx && action == 4) action = 6;            // Env_transact.env_action
            if (V4) {
                if (has_draws) {
                    const uint32_t *dr = p.draws + idx * TW_DRAW_WORDS;
                    dw[0] = dr[0]; dw[1] = dr[1]; dw[2] = dr[2]; dw[3] = dr[3];
                } else {
                    draw_block(p.seed_lo, p.seed_hi, env_id, t_now, 0, dw);
                }
            }
            if (action >= 7) action = 0;                          // twoarmy_v6.py:85-86
        }
        // TURBO step: every env of the wave is in normal play (mode_ok: adjacent ball triple on row 8 well
        // inside the grid, facing up, agent inside the border -> nothing can raise) and got a legal action
        // {left,right,up,down,done}.  Then the whole pre-observation transition is branch-free.
        const bool legal = ((unsigned)action <= 6u) & (((0x4Fu >> (action & 7)) & 1u) != 0u);
        const bool all_turbo = __ballot(active && !(mode_ok && legal)) == 0ull;
        if (all_turbo) {
            if (active) {
                s.t += 1;
                s.step_move += 1;                                 // twoarmy_v6.py:88
                s.m6 = s.m6 == 5 ? 0 : s.m6 + 1;
                s.m4 = (s.m4 + 1) & 3;
                const int m6 = s.m6;
                const int dxb = (m6 <= 1) ? 1 : (m6 <= 3 ? -1 : 0);   // :104-109
                // row-8 balls (:96-112) as 4 branch-free cell writes
                uint32_t *row8 = my_env + gpi(0, 8);      // row 8 of the x-major image: stride GPP per x
                uint32_t *mrow8 = my_env + MAT_OFF + 8 * GS;
                const int b0 = s.obx[0], nb = b0 + dxb;
                const int cx = dxb > 0 ? b0 : b0 + 2;             // the vacated cell (a ball cell when dxb == 0)
                uint8_t *crow8 = mcb(my_env) + 8 * GS;
                row8[cx * GPP] = dxb != 0 ? C_EMPTY : C_BALL;
                mrow8[cx] = dxb != 0 ? M_FREE : M_BALL;
                crow8[cx] = dxb != 0 ? 0 : 2;
                row8[nb * GPP] = C_BALL; row8[(nb + 1) * GPP] = C_BALL; row8[(nb + 2) * GPP] = C_BALL;
                mrow8[nb] = M_BALL; mrow8[nb + 1] = M_BALL; mrow8[nb + 2] = M_BALL;
                crow8[nb] = 2; crow8[nb + 1] = 2; crow8[nb + 2] = 2;
                s.obx[0] = nb; s.obx[1] = nb + 1; s.obx[2] = nb + 2;
                bool alive = true;
                if (V4) {
                    if (s.upd_long) {                             // twoarmy_v4.py:115-144
                        s.upd_horiz = 0;
                        bool go = (s.m4 == 2) || (m6 == 3) || (m6 == 0);
                        if (!go) go = (dw[TW_S_GATE] % 10u) == 6u;
                        if (go && s.patrol) {
                            if (s.up1) {
                                alive = move_group<3>(my_env, s.o1x, s.o1y, s.o1v, 0, -1, err);
                                if (alive && s.o1y[0] == 3) s.up1 = 0;
                            } else {
                                alive = move_group<3>(my_env, s.o1x, s.o1y, s.o1v, 0, 1, err);
                                if (alive && s.o1y[2] == 7) s.up1 = 1;
                            }
                        }
                    }
                    if (alive && s.upd_horiz) {                   // twoarmy_v4.py:147-176
                        s.upd_long = 0;
                        bool go = (m6 != 1);
                        if (!go) go = (dw[TW_S_GATE] % 10u) == 6u;
                        if (go && s.patrol) {
                            if (s.right2) {
                                alive = move_group<4>(my_env, s.o2x, s.o2y, s.o2v, 1, 0, err);
                                if (alive && s.o2x[3] == 11) s.right2 = 0;
                            } else {
                                alive = move_group<4>(my_env, s.o2x, s.o2y, s.o2v, -1, 0, err);
                                if (alive && s.o2x[0] == 5) s.right2 = 1;
                            }
                        }
                    }
                }
                if (alive) {
                    // MiniGridEnv.step (minigrid.py:1333-1441): dx/dy packed as 2-bit fields (value+1) per action
                    s.step_count += 1;
                    const int tx = s.ax + (int)((0x1558u >> (2 * action)) & 3u) - 1;     // L,R,U,D,-,-,stay
                    const int ty = s.ay + (int)((0x1585u >> (2 * action)) & 3u) - 1;
                    const uint32_t cv = my_env[gpi(tx, ty)];
                    const uint32_t ct = cv & 0xffu, cs = (cv >> 16) & 0xffu;
                    const bool enter = ((ct < 12u) & ((0x0B0Au >> (ct & 15u)) & 1u)) | ((ct == 4u) & (cs == 0u));
                    s.ax = enter ? tx : s.ax;
                    s.ay = enter ? ty : s.ay;
                    terminated = ct == 8u;
                    truncated = s.step_count >= s.max_steps;      // :1436-1437
                    have_obs = 1;
                    // keep the invariants of the turbo step for the next one
                    mode_ok = ((unsigned)(nb - 2) <= 10u) & ((unsigned)(s.ax - 1) <= 14u) & ((unsigned)(s.ay - 1) <= 14u);
                } else {
                    mode_ok = false;
                }
            }
        } else if (active) {
            // GENERAL step: literal transition with every bounds check / raise of the reference
            s.t += 1;
            s.step_move += 1;                                     // :88
            s.m6 = s.m6 == 5 ? 0 : s.m6 + 1;
            s.m4 = (s.m4 + 1) & 3;
            const int m6 = s.m6;
            const int dxb = (m6 <= 1) ? 1 : (m6 <= 3 ? -1 : 0);   // :104-109
            bool alive = true;
            // ---- row-8 balls (:96-112): clear all three, then put each inside try/except.
            // Fast path (always taken in normal play): the balls are the adjacent triple b..b+2 on row 8
            // and the moved triple stays inside the grid -> 4 branch-free cell writes.
            const int b0 = s.obx[0];
            const bool balls_fast = (s.oby[0] == 8) & (s.oby[1] == 8) & (s.oby[2] == 8) & (s.obx[1] == b0 + 1) &
                                    (s.obx[2] == b0 + 2) & ((unsigned)(b0 + dxb - 1) <= 12u) & ((unsigned)(b0 - 1) <= 12u);
            if (balls_fast) {
                uint32_t *row8 = my_env + gpi(0, 8);      // row 8 of the x-major image: stride GPP per x
                uint32_t *mrow8 = my_env + MAT_OFF + 8 * GS;
                const int nb = b0 + dxb;
                const int cx = dxb > 0 ? b0 : b0 + 2;             // the vacated cell (a ball cell when dxb == 0)
                uint8_t *crow8 = mcb(my_env) + 8 * GS;
                row8[cx * GPP] = dxb != 0 ? C_EMPTY : C_BALL;
                mrow8[cx] = dxb != 0 ? M_FREE : M_BALL;
                crow8[cx] = dxb != 0 ? 0 : 2;
                row8[nb * GPP] = C_BALL; row8[(nb + 1) * GPP] = C_BALL; row8[(nb + 2) * GPP] = C_BALL;
                mrow8[nb] = M_BALL; mrow8[nb + 1] = M_BALL; mrow8[nb + 2] = M_BALL;
                crow8[nb] = 2; crow8[nb + 1] = 2; crow8[nb + 2] = 2;
                s.obx[0] = nb; s.obx[1] = nb + 1; s.obx[2] = nb + 2;
            } else {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (ok) { if (inb(s.obx[k], s.oby[k])) put(my_env, s.obx[k], s.oby[k], C_EMPTY, M_FREE); else ok = false; }
                if (!ok) { err = TW_ENV_ASSERT; alive = false; }
                else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int nx = s.obx[k] + dxb;
                        if (inb(nx, 8)) { put(my_env, nx, 8, C_BALL, M_BALL); s.obx[k] = nx; s.oby[k] = 8; }
                    }
                }
            }
            if (V4 && alive) {
                if (s.upd_long) {                                 // twoarmy_v4.py:115-144
                    s.upd_horiz = 0;
                    bool go = (s.m4 == 2) || (m6 == 3) || (m6 == 0);
                    if (!go) go = (dw[TW_S_GATE] % 10u) == 6u;
                    if (go && s.patrol) {
                        if (s.up1) {
                            alive = move_group<3>(my_env, s.o1x, s.o1y, s.o1v, 0, -1, err);
                            if (alive && s.o1y[0] == 3) s.up1 = 0;
                        } else {
                            alive = move_group<3>(my_env, s.o1x, s.o1y, s.o1v, 0, 1, err);
                            if (alive && s.o1y[2] == 7) s.up1 = 1;
                        }
                    }
                }
                if (alive && s.upd_horiz) {                       // twoarmy_v4.py:147-176
                    s.upd_long = 0;
                    bool go = (m6 != 1);
                    if (!go) go = (dw[TW_S_GATE] % 10u) == 6u;
                    if (go && s.patrol) {
                        if (s.right2) {
                            alive = move_group<4>(my_env, s.o2x, s.o2y, s.o2v, 1, 0, err);
                            if (alive && s.o2x[3] == 11) s.right2 = 0;
                        } else {
                            alive = move_group<4>(my_env, s.o2x, s.o2y, s.o2v, -1, 0, err);
                            if (alive && s.o2x[0] == 5) s.right2 = 1;
                        }
                    }
                }
            }
            if (alive) {
                // ---- MiniGridEnv.step (minigrid.py:1333-1441)
                s.step_count += 1;
                const int tx = s.ax + (action == 1) - (action == 0);
                const int ty = s.ay + (action == 3) - (action == 2);
                // fast path: legal action, facing up, agent strictly inside the border -> front_pos and the
                // target cell are in bounds, nothing can raise
                const bool move_fast = (((unsigned)action <= 3u) | (action == 6)) & (s.dir == 3) &
                                       ((unsigned)(s.ax - 1) <= 14u) & ((unsigned)(s.ay - 1) <= 14u);
                bool legal = move_fast;
                if (!move_fast) {
                    const int fx = s.ax + (s.dir == 0 ? 1 : (s.dir == 2 ? -1 : 0));   // front_pos (:1341-1344)
                    const int fy = s.ay + (s.dir == 1 ? 1 : (s.dir == 3 ? -1 : 0));
                    if (!inb(fx, fy)) err = TW_ENV_ASSERT;
                    else if (action > 3 && action != 6) err = TW_ENV_ATTRIBUTE;   // self.actions.forward, :1397
                    else if (action < 0) err = TW_ENV_ATTRIBUTE;
                    else if (!inb(tx, ty)) err = TW_ENV_ASSERT;
                    else legal = true;
                }
                if (legal) {
                    const uint32_t cv = my_env[gpi(tx, ty)];
                    const uint32_t ct = cv & 0xffu, cs = (cv >> 16) & 0xffu;
                    // empty(1) or can_overlap: floor 3, goal 8, lava 9, subgoal 11, open door (4, state 0)
                    const bool enter = ((ct < 12u) & ((0x0B0Au >> (ct & 15u)) & 1u)) | ((ct == 4u) & (cs == 0u));
                    s.ax = enter ? tx : s.ax;
                    s.ay = enter ? ty : s.ay;
                    terminated = ct == 8u;
                    truncated = s.step_count >= s.max_steps;      // :1436-1437
                    have_obs = 1;
                }
            }
            mode_ok = have_obs && normal_mode(s);
        }
        if (active && !have_obs) {     // the reference raised: state keeps the mutations made so far
            s.err = err; s.last_reward = -1; s.last_term = 0; s.last_trunc = 0;
        }
        wave_sync();

        // ================= LOGIC part 2 (lane-per-env): everything after gen_obs().
        // CELLS=false is the common case in which no lane drops walls / spawns patrols / raises this
        // step, so part 2 touches no LDS cell and may be scheduled among the emission loads.
        int done = 0;
        auto part2 = [&](auto cells_tag) {
            constexpr bool CELLS = decltype(cells_tag)::value;
            if (!(active && have_obs)) return;
            if (CELLS) {
                if (!s.pone && (s.ax > 3 || s.ay < 14)) {         // twoarmy_v6.py:182-198 / v4:181-195
                    int i1 = 11, i2 = 8;
                    if (V4) { i1 = 9 + (int)(dw[TW_S_WALL1] % 4u); i2 = 6 + (int)(dw[TW_S_WALL2] % 4u); }
                    put(my_env, 4, i1, C_WALL, M_WALL); put(my_env, 5, i1, C_WALL, M_WALL);
                    put(my_env, 4, i1 + 1, C_WALL, M_WALL); put(my_env, 5, i1 + 1, C_WALL, M_WALL);
                    put(my_env, i2, 11, C_WALL, M_WALL); put(my_env, i2, 12, C_WALL, M_WALL);
                    put(my_env, i2 + 1, 11, C_WALL, M_WALL); put(my_env, i2 + 1, 12, C_WALL, M_WALL);
                    s.pone = 1; s.wall_i1 = i1; s.wall_i2 = i2;
                }
                if (V4 && !s.patrol && s.ay <= 8) {               // twoarmy_v4.py:212-225
                    const int i = 6 + (int)(dw[TW_S_SPAWN] % 4u);
                    s.o2x[0] = i; s.o2x[1] = i + 1; s.o2x[2] = i; s.o2x[3] = i + 1;
                    s.o2y[0] = 4; s.o2y[1] = 4; s.o2y[2] = 5; s.o2y[3] = 5;
#pragma unroll
                    for (int k = 0; k < 4; ++k) put(my_env, s.o2x[k], s.o2y[k], C_BALL, M_BALL);
#pragma unroll
                    for (int k = 0; k < 3; ++k) { s.o1x[k] = 12; s.o1y[k] = 4 + k; put(my_env, 12, 4 + k, C_BALL, M_BALL); }
                    s.o1v = 1; s.o2v = 1; s.patrol = 1;
                }
            }
            // row-ball collision / proximity (twoarmy_v6.py:231-243), branch-free
            bool hit = false, risk = false;
            if (all_turbo) {                                      // adjacent triple on row 8: one span test
                const bool in_span = (unsigned)(s.ax - s.obx[0]) <= 2u;
                hit = in_span & (s.ay == 8);
                risk = in_span & (s.ay == 9);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) hit |= (s.ax == s.obx[k]) & (s.ay == s.oby[k]);
                risk = (s.ay == s.oby[0] + 1) & ((s.ax == s.obx[0]) | (s.ax == s.obx[1]) | (s.ax == s.obx[2]));
            }
            reward = hit ? R_HIT : reward;
            reward = risk ? R_RISK : reward;
            if (s.patrol) {                                       // :245-283 (dead in v6: patrol is never set)
                if (CELLS && (!s.o1v || !s.o2v)) err = TW_ENV_TYPE;
                else {
                    bool prisk = (s.ay == s.o2y[2] + 1) & ((s.ax == s.o2x[2]) | (s.ax == s.o2x[3]));
                    prisk |= (s.ax == s.o2x[0] - 1) & ((s.ay == s.o2y[0]) | (s.ay == s.o2y[2]));
                    prisk |= (s.ax == s.o2x[1] + 1) & ((s.ay == s.o2y[1]) | (s.ay == s.o2y[3]));
                    prisk |= (s.ax == s.o1x[0] - 1) & ((s.ay == s.o1y[0]) | (s.ay == s.o1y[1]) | (s.ay == s.o1y[2]));
                    bool phit = false;
#pragma unroll
                    for (int k = 0; k < 3; ++k) phit |= (s.ax == s.o1x[k]) & (s.ay == s.o1y[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) phit |= (s.ax == s.o2x[k]) & (s.ay == s.o2y[k]);
                    reward = prisk ? R_RISK : reward;
                    reward = phit ? R_HIT : reward;
                    hit |= phit;
                }
            }
            truncated |= hit;
            if (!CELLS || err == TW_ENV_OK) {
                const bool room2 = s.first_room2 & (s.ay == 7);   // :285-288
                reward = room2 ? R_ROOM2 : reward;
                s.first_room2 = room2 ? 0 : s.first_room2;
                s.risk += (reward == R_RISK);                     // :290-294
                truncated |= (reward == R_RISK) & (s.risk > 5);
                if (terminated || truncated) {                    // :296-318
                    if (terminated) reward = R_GOAL;
                    s.step_move = 0; s.m6 = 0; s.m4 = 0;
                    s.pone = 0; s.patrol = 0; s.first_room2 = 1; s.risk = 0;
                    uint32_t cw[4];
                    if (has_draws) {
                        const uint32_t *dr = p.draws + idx * TW_DRAW_WORDS;
                        cw[0] = dr[TW_S_COIN_A]; cw[1] = dr[TW_S_COIN_B];
                    } else {
                        draw_block(p.seed_lo, p.seed_hi, env_id, t_now, 1, cw);
                    }
                    if ((cw[0] & 1u) == 1u) { s.up1 = 0; s.right2 = 1; } else { s.up1 = 1; s.right2 = 0; }
                    if ((cw[1] & 1u) == 1u) { s.upd_horiz = 0; s.upd_long = 1; } else { s.upd_horiz = 1; s.upd_long = 0; }
                    s.episodes += 1;
                    done = 1;
                }
                s.last_reward = reward; s.last_term = terminated; s.last_trunc = truncated;
                if (FAST || p.reward) p.reward[idx] = reward_value(reward);
                if (FAST || p.term) p.term[idx] = (uint8_t)terminated;
                if (FAST || p.trunc) p.trunc[idx] = (uint8_t)truncated;
                if (FAST || p.pos) { float2 ps; ps.x = (float)s.ay; ps.y = (float)s.ax; reinterpret_cast<float2 *>(p.pos)[idx] = ps; }
            } else {
                s.last_reward = -1; s.last_term = 0; s.last_trunc = 0;
                have_obs = 0;                                     // no post-step outputs for a raised step
            }
            s.err = err;
        };

        STAMP(2);
        // ---- wave-uniform path selection
        const bool cells2 = active && have_obs &&
                            ((!s.pone && (s.ax > 3 || s.ay < 14)) || (V4 && !s.patrol && s.ay <= 8) ||
                             (s.patrol && (!s.o1v || !s.o2v)));
        const bool common = layouts_fast && __ballot(active && have_obs && s.dir == 3) == m_active &&
                            __ballot(cells2) == 0ull;
        if (common) {
            // ================= COMMON PATH: all gathers first, part 2 in their shadow, then pack + store
            uint32_t oc[E][6];
            uint4 mq[E][2];
            int cax[E], cay[E];
            const int qb = lane < 9 ? 64 + lane : 72;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                cax[e] = __builtin_amdgcn_readlane(s.ax, e);
                cay[e] = __builtin_amdgcn_readlane(s.ay, e);
                const uint32_t *env = lds_env + e * ENV_WORDS;
                const char *base = reinterpret_cast<const char *>(env + gpi(cax[e], cay[e]));
                if (has_obs) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) oc[e][k] = *reinterpret_cast<const uint32_t *>(base + of.rel4[k]);
                }
                if (has_mat && !code_mode) {
                    mq[e][0] = *reinterpret_cast<const uint4 *>(env + MAT_OFF + 4 * lane);
                    mq[e][1] = *reinterpret_cast<const uint4 *>(env + MAT_OFF + 4 * qb);
                }
            }
            STAMP(3);
            part2(std::false_type{});
            STAMP(4);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool valid_e = n0 + e < N;
                if (has_obs) {
                    const uint32_t *c = oc[e];
                    const uint32_t w0 = c[0] | (c[1] << 24);
                    const uint32_t w1 = (c[1] >> 8) | (c[2] << 16);
                    const uint32_t w2 = (c[2] >> 16) | (c[3] << 8);
                    const uint32_t w3 = c[4] | (c[5] << 24);
                    const uint32_t w4 = c[5] >> 8;
                    uint4 o;
                    o.x = (__builtin_amdgcn_alignbit(w1, w0, of.shift) & of.andm[0]) | of.orm[0];
                    o.y = (__builtin_amdgcn_alignbit(w2, w1, of.shift) & of.andm[1]) | of.orm[1];
                    o.z = (__builtin_amdgcn_alignbit(w3, w2, of.shift) & of.andm[2]) | of.orm[2];
                    o.w = (__builtin_amdgcn_alignbit(w4, w3, of.shift) & of.andm[3]) | of.orm[3];
                    if (of.active && valid_e)
                        *reinterpret_cast<uint4 *>(obs_row + (size_t)e * p.obs_pitch + 16 * lane) = o;
                }
                if (has_mat && code_mode) {
                    if (valid_e) emit_codes(lds_env + e * ENV_WORDS, cax[e], cay[e], lane, matc_row + (size_t)e * p.mat_pitch, fast_code_ok);
                } else if (has_mat) {
                    const int ca = cay[e] * GS + cax[e];
                    float *dst = mat_row + (size_t)e * p.mat_pitch;
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const int q = pass == 0 ? lane : qb;
                        uint4 v = mq[e][pass];
                        const int d = ca - 4 * q;
                        v.x = d == 0 ? M_AGENT : v.x;
                        v.y = d == 1 ? M_AGENT : v.y;
                        v.z = d == 2 ? M_AGENT : v.z;
                        v.w = d == 3 ? M_AGENT : v.w;
                        if (valid_e && (pass == 0 || lane < 9)) *reinterpret_cast<uint4 *>(dst + 4 * q) = v;
                    }
                }
            }
        } else {
            // ================= ORDERED PATH: obs -> part 2 (cell writes) -> matrix, per-env generic fallbacks
            if (has_obs) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (!__builtin_amdgcn_readlane(have_obs, e)) continue;
                    const int ax = __builtin_amdgcn_readlane(s.ax, e), ay = __builtin_amdgcn_readlane(s.ay, e);
                    const int dir = __builtin_amdgcn_readlane(s.dir, e);
                    const uint32_t *env = lds_env + e * ENV_WORDS;
                    uint8_t *dst = obs_row + (size_t)e * p.obs_pitch;
                    if (fast_obs_ok && dir == 3) emit_obs_fast(env, ax, ay, of, lane, dst);
                    else emit_obs_generic([env](int x, int y) { return env[gpi(x, y)]; }, stage, lane, V, ax, ay, dir, dst);
                }
            }
            part2(std::true_type{});
            wave_sync();
            if (has_mat) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (!__builtin_amdgcn_readlane(have_obs, e)) continue;
                    const int ax = __builtin_amdgcn_readlane(s.ax, e), ay = __builtin_amdgcn_readlane(s.ay, e);
                    uint32_t *env = lds_env + e * ENV_WORDS;
                    float *dst = mat_row + (size_t)e * p.mat_pitch;
                    if (code_mode) emit_codes(env, ax, ay, lane, matc_row + (size_t)e * p.mat_pitch, fast_code_ok);
                    else if (fast_mat_ok) emit_matrix_fast(env, ax, ay, lane, dst);
                    else emit_matrix_generic(env, ax, ay, lane, dst);
                }
            }
        }

        STAMP(5);
        // ================= auto-reset (soa/train_ppo.py:104): ballot of done envs, regenerate in place
        if (autoreset) {
            const unsigned long long dm = __ballot(done != 0);
            if (dm) {
                wave_sync();
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((dm >> e) & 1ull) regen_env(lds_env + e * ENV_WORDS, lane);
                if (done) { reset_scalars(s); mode_ok = true; }
                wave_sync();
            }
        }
    }

#ifdef TW_STAMP
    STAMP(6);
    if (lane == 0 && blockIdx.x < 64) for (int i = 0; i < 8; ++i) g_stamp[blockIdx.x][i] = st_acc[i];
#endif
    // ---- write state back
    wave_sync();
    if (lane < E) store_env(s, recs + lane * REC);
    wave_sync();
    for (int i = lane; i < E * REC; i += 64) {
        const int e = i / REC;
        if (n0 + e < N) p.rec_out[(size_t)n0 * REC + i] = recs[i];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (n0 + e >= N) break;
        const uint32_t *env = lds_env + e * ENV_WORDS;
        const size_t gb = (size_t)(n0 + e) * NC;
        for (int c = lane; c < NC; c += 64) {
            const int y = c / GS, x = c - y * GS;
            const uint32_t v = env[gpi(x, y)];
            p.type_out[gb + c] = (uint8_t)v;
            p.colour_out[gb + c] = (uint8_t)(v >> 8);
        }
    }
}

// One wavefront per workgroup; workgroup b owns envs b*E .. b*E+E-1.  The flag-gated fallback launch behind the pipelined
// kernel uses a SMALL grid and walks the env groups with a grid stride: when the flag is clear (always, in normal play)
// its cost is one kernel boundary instead of the dispatch of N/E workgroups (6 us -> ~2 us per rollout at 4096 envs).
template <int E, int VARIANT, bool FAST>
__global__ __launch_bounds__(64, (E == 1 ? 4 : 2)) void tw_rollout_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_env[E * ENV_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t stage[STAGE_WORDS];
    __shared__ int32_t recs[E * REC];
    __shared__ int32_t act_lds[64 * E];
    if (p.only_if_flagged) {                             // fallback launch behind the pipelined kernel
        if (*p.abnormal == 0) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(p.fb_count, 1);
    }
    const int groups = (p.n_envs + E - 1) / E;
    for (int b = blockIdx.x; b < groups; b += gridDim.x) {
        rollout_body<E, VARIANT, FAST>(p, b * E, threadIdx.x, lds_env, stage, recs, act_lds);
        wave_sync();
    }
}


// ---------------------------------------------------------------- the pipelined rollout kernel
// Normal play only (auto-reset on, Philox draws, native layouts, facing up): in that regime the grid is
// a closed-form function of seven small integers,
//     static map + ball triple at b0 + [pone: two 2x2 wall blocks at i1 / i2]
//                + [patrol: column of 3 balls at (12, o1y0..) and 2x2 balls at (o2x0.., 4..5)],
// so the sequential part of a step (the transition) and the heavy part (emitting 2 KB of observation +
// state matrix) can be decoupled.  One workgroup = 16 waves owns 16 envs:
//   * wave 0 (LOGIC): lanes 0..15 run the transition analytically -- no LDS grid -- and publish one packed
//     32-bit record per env-step into an LDS ring, plus reward / terminated / truncated / pos to HBM;
//   * all waves (EMIT, wave 0 joins when its chunk is done) pull (step, env) tasks from an LDS counter,
//     patch <= 18 dynamic cells into their private copy of the static wall-padded grid + matrix image,
//     emit with the same register-packed paths as the sequential kernel, and un-patch.
// Emission of different steps is independent, so the 15 emit waves hide each other's latencies and the
// kernel runs at the HBM write rate instead of at one wave's dependent-instruction latency.
// Anything outside normal play (illegal action, drifted balls, injected grids, ...) raises *p.abnormal;
// the host always enqueues the sequential kernel behind this one, which re-runs the launch from the
// untouched input state iff the flag is set (both write the `_out` state; the host swaps afterwards).
#ifndef TW_PWAVES
#define TW_PWAVES 16
#endif
#ifndef TW_WG_TARGET
#define TW_WG_TARGET 256
#endif
constexpr int PWAVES = TW_PWAVES;   // waves per workgroup (wave e verifies env e of the group: PG <= PWAVES)
constexpr int PG_MAX = PWAVES < 16 ? PWAVES : 16;   // envs per workgroup (template parameter PG = 16, 8, 4 or 2: small batches still fill the CUs)
constexpr int PIPE_WG_TARGET = TW_WG_TARGET;        // envs per workgroup are chosen so that about this many workgroups exist
constexpr int PCH = 128;        // steps per ring chunk
// env-steps one emission wave takes per draw from the task counter: a compile-time choice per frame layout.
// Float frames: 1 -- the 16 waves of a workgroup then write 16 neighbouring rows of the same step at any time; with 8 (one
// wave streaming 16 KB on its own) the same stores ran 5-17 % slower on every box tried, with 2 still 5 % slower, and a
// store-only replica of the pattern (tools/store_pattern_probe.hip) drops from ~6.3 to ~4.3 TB/s: that layout is bound by
// the store path, not by issue.  Code frames (TW_F_MATRIX_CODE, 1184-byte records) are issue-bound instead: 4 tasks per
// draw save three of four counter / poll / index sequences (6 % per launch, tools/ab_variants.py with AB_CODES=1), and the
// code-frame instantiation of the kernel drops every float-matrix path at compile time.
#ifdef TW_PGRP
constexpr int PGRP_FLOAT = TW_PGRP, PGRP_CODE = TW_PGRP;      // diagnostic builds: one value for both layouts
#else
constexpr int PGRP_FLOAT = 1, PGRP_CODE = 4;
#endif
constexpr uint32_t REC_VALID = 0x80000000u;

struct Dyn { int b0, pone, i1, i2, patrol, o1y0, o2x0; };
// rare events of the logic loop (wall drop, patrol spawn, episode end): their code is laid out behind the loop so that the
// common path falls through (a taken branch costs the lone logic wave an instruction-buffer refill)
#ifdef LG_NOEXPECT
#define LG_RARE(c) (c)
#else
#define LG_RARE(c) __builtin_expect(!!(c), 0)
#endif
#ifndef LG_UNROLL
#define LG_UNROLL 1
#endif
// (x << K) | acc in ONE instruction; written as plain C the optimiser re-balances a chain of these into separate shifts
// plus 3-input ORs (17 instead of 7 instructions for the record word)
template <int K>
__device__ __forceinline__ uint32_t lshl_or(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(K), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t static_cell(int x, int y) {      // _gen_grid without the balls
    if (x == 0 || y == 0 || x == GS - 1 || y == GS - 1) return C_WALL;
    if (y == 8) return (x <= 5 || x >= 11) ? C_WALL : C_EMPTY;
    if (x == 14 && y == 2) return C_GOAL;
    return C_EMPTY;
}

__device__ __forceinline__ uint32_t analytic_cell(int x, int y, const Dyn &d) {
    uint32_t c = static_cell(x, y);
    if (y == 8 && (unsigned)(x - d.b0) <= 2u) c = C_BALL;
    if (d.pone && ((((unsigned)(x - 4) <= 1u) & ((unsigned)(y - d.i1) <= 1u)) |
                   (((unsigned)(x - d.i2) <= 1u) & ((unsigned)(y - 11) <= 1u)))) c = C_WALL;
    if (d.patrol && (((x == 12) & ((unsigned)(y - d.o1y0) <= 2u)) |
                     (((unsigned)(x - d.o2x0) <= 1u) & ((unsigned)(y - 4) <= 1u)))) c = C_BALL;
    return c;
}

// is the scalar state inside the closed-form regime?  (the planes are checked against analytic_cell separately)
template <bool V4>
__device__ __forceinline__ bool pipe_state_ok(const EnvS &s) {
    bool ok = normal_mode(s) & ((unsigned)(s.obx[0] - 6) <= 2u) & (s.err == 0) & (s.gx == 14) & (s.gy == 2) &
              (s.max_steps > 0);
    ok &= s.obx[0] == (int)((0x666787u >> (4 * ((uint32_t)s.step_move % 6u))) & 15u);   // the logic wave's closed form
    if (s.pone) ok &= ((unsigned)(s.wall_i1 - 9) <= 3u) & ((unsigned)(s.wall_i2 - 6) <= 3u);
    if (V4) {
        if (s.patrol) {
            ok &= (s.o1v != 0) & (s.o2v != 0) & (s.o1x[0] == 12) & (s.o1x[1] == 12) & (s.o1x[2] == 12) &
                  (s.o1y[1] == s.o1y[0] + 1) & (s.o1y[2] == s.o1y[0] + 2) & ((unsigned)(s.o1y[0] - 3) <= 2u) &
                  ((unsigned)(s.o2x[0] - 5) <= 5u) & (s.o2x[1] == s.o2x[0] + 1) & (s.o2x[2] == s.o2x[0]) &
                  (s.o2x[3] == s.o2x[0] + 1) & (s.o2y[0] == 4) & (s.o2y[1] == 4) & (s.o2y[2] == 5) & (s.o2y[3] == 5);
        }
    } else {
        ok &= (s.patrol == 0);
    }
    return ok;
}

// 31-bit record of one env-step (bit 31 = valid): everything the emission waves need, including the scalar outputs
//   0-4 ax  5-9 ay  10-11 b0-6  12-13 o1y0-3  14-16 o2x0-5  17-18 i1-9  19-20 i2-6
//   21 pone(obs)  22 pone(matrix)  23 patrol(obs)  24 patrol(matrix)  25-27 reward code  28 terminated  29 truncated
__device__ __forceinline__ uint32_t pack_record(int ax, int ay, const Dyn &d, int pone_pre, int patrol_pre, int reward,
                                                int terminated, int truncated) {
    return (uint32_t)ax | ((uint32_t)ay << 5) | ((uint32_t)((d.b0 - 6) & 3) << 10) | ((uint32_t)((d.o1y0 - 3) & 3) << 12) |
           ((uint32_t)((d.o2x0 - 5) & 7) << 14) | ((uint32_t)((d.i1 - 9) & 3) << 17) | ((uint32_t)((d.i2 - 6) & 3) << 19) |
           ((uint32_t)pone_pre << 21) | ((uint32_t)d.pone << 22) | ((uint32_t)patrol_pre << 23) |
           ((uint32_t)d.patrol << 24) | ((uint32_t)reward << 25) | ((uint32_t)(terminated != 0) << 28) |
           ((uint32_t)(truncated != 0) << 29);
}

// Per-engine constant tables of the pipelined kernel, built once at tw_create: what every workgroup used to recompute
// in its prologue (1155 + 292 + 76 image words per wave with index divisions, and make_obs_fast's divisions by the
// runtime view size) -- at 1024 envs per GPU the prologue was 21 % of the kernel (30 of 142 k cycles), at 4096 still 10 %.
//   [PT_IMG, +ENV_WORDS)   the static wall-padded code image | float matrix image | matrix code bytes of an empty map
//   [PT_OBS, +2*64*16)     ObsFast of lane l for the two-stream layout (chunk = l) and the record layout (chunk = l - 9):
//                          words 0-5 rel4, 6 shift, 7-10 andm, 11-14 orm, 15 active
constexpr int PT_IMG = 0, PT_OBS = ENV_WORDS, PT_WORDS = PT_OBS + 2 * 64 * 16;
static_assert(ENV_WORDS % 4 == 0 && ENV_WORDS / 4 <= 6 * 64, "image copied as 6 x dwordx4 per lane");

__global__ __launch_bounds__(64) void tw_pipe_tables_kernel(uint32_t *tab, int V) {
    const int lane = threadIdx.x;
    uint32_t *im = tab + PT_IMG;
    for (int i = lane; i < ENV_WORDS; i += 64) im[i] = 0u;
    __syncthreads();
    for (int i = lane; i < GP_WORDS; i += 64) {
        const int x = i / GPP - GPX0, y = i - (i / GPP) * GPP - GPY0;
        im[i] = inb(x, y) ? static_cell(x, y) : C_WALL;
    }
    for (int c = lane; c < MAT_WORDS; c += 64)
        im[MAT_OFF + c] = c < NC ? mat_of_code(static_cell(c - (c / GS) * GS, c / GS)) : 0u;
    for (int c = lane; c < MATC_BYTES; c += 64)
        mcb(im)[c] = c < NC ? mcode_of(mat_of_code(static_cell(c - (c / GS) * GS, c / GS))) : 0;
    for (int r = 0; r < 2; ++r) {
        const ObsFast f = make_obs_fast(r ? lane - 9 : lane, V);
        uint32_t *o = tab + PT_OBS + (r * 64 + lane) * 16;
        for (int k = 0; k < 6; ++k) o[k] = (uint32_t)f.rel4[k];
        o[6] = f.shift;
        for (int k = 0; k < 4; ++k) { o[7 + k] = f.andm[k]; o[11 + k] = f.orm[k]; }
        o[15] = (uint32_t)f.active;
    }
}

// LAYOUT: 0 two float streams (obs rows, matrix rows), 1 one stream of float records (tw_alloc_outputs), 2 code frames
template <int VARIANT, int PG, int LAYOUT>
__global__ __launch_bounds__(64 * PWAVES, 16 / PWAVES) void tw_pipe_kernel(Params p) {
    constexpr bool CODE = LAYOUT == 2;
    constexpr int PGRP = (CODE ? PGRP_CODE : PGRP_FLOAT) < PG ? (CODE ? PGRP_CODE : PGRP_FLOAT) : PG;
    extern __shared__ __attribute__((aligned(16))) uint32_t pipe_lds[];
    uint32_t *img = pipe_lds;                                   // [PWAVES][ENV_WORDS]
    uint32_t *ring = img + PWAVES * ENV_WORDS;                  // [PCH][PG]
    int32_t *recs = reinterpret_cast<int32_t *>(ring + PCH * PG);   // [PG][REC]
    volatile int *ctrl = reinterpret_cast<volatile int *>(recs + PG * REC);   // [0] ready  [1] next task
    uint32_t *drw = const_cast<uint32_t *>(reinterpret_cast<volatile uint32_t *>(ctrl + 4));  // [PCH][PG] packed draws + action

    constexpr bool V4 = VARIANT == 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.n_envs;
#ifdef TW_STAMP
    unsigned long long pst_entry = 0, pst_lstart = 0, pst_lend = 0, pst_p1 = 0, pst_p2 = 0, pst_p3 = 0, pst_p4 = 0;
#define PSTAMP0(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_entry) :: "memory");
#endif
    // workgroups are dealt round-robin to the 8 XCDs: give each XCD one contiguous range of envs so that the
    // partially written lines of the [T][N] scalar outputs (reward / terminated / truncated / pos) merge in ONE L2
    const int n0 = ((gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x) * PG;
    uint32_t *my_img = img + wave * ENV_WORDS;

    // ---- every global load of the prologue is issued up front and UNCONDITIONALLY (clamped indices instead of
    // predicates: a predicated load becomes a branch with its own wait) -- ONE HBM round trip instead of three:
    //   * the static image goes from the per-engine table straight into this wave's LDS copy (LDS-direct loads, no VGPRs;
    //     lane l of chunk k lands at my_img + 16 (64 k + l) bytes),
    //   * this lane's emission constants (same table), the block's records, the planes to verify, the first chunk's actions.
    constexpr bool record = LAYOUT == 1;                        // == (p.record != 0): the host picks the instantiation
    const uint4 *tab4 = reinterpret_cast<const uint4 *>(p.pipe_tab);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (lane + 64 * k < ENV_WORDS / 4)
            __builtin_amdgcn_global_load_lds(tab4 + lane + 64 * k,
                                             (__attribute__((address_space(3))) void *)(my_img + 256 * k), 16, 0, 0);
    const uint4 *of4 = tab4 + PT_OBS / 4 + ((record ? 64 : 0) + lane) * 4;
    const uint4 ofw0 = of4[0], ofw1 = of4[1], ofw2 = of4[2], ofw3 = of4[3];
    const bool own_env = wave < PG && n0 + wave < N;              // wave e loads and verifies env n0 + e
    const int ve = min(n0 + (wave < PG ? wave : 0), N - 1);
    const int32_t rec_pre = p.rec[(size_t)ve * REC + (lane < REC ? lane : 0)];
    uint32_t ty_pre[5], co_pre[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const size_t g = (size_t)ve * NC + min(lane + 64 * k, NC - 1);
        ty_pre[k] = p.type[g]; co_pre[k] = p.colour[g];
    }
    constexpr int NPRE = (PCH * PG + 64 * PWAVES - 1) / (64 * PWAVES);   // staging items per thread and chunk
    int act_pre[NPRE];
    {
        const int len0 = min(PCH, p.T);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * 64 * PWAVES, tl = i / PG, e = i - tl * PG;
            act_pre[k] = p.actions[(size_t)min(tl, len0 - 1) * N + min(n0 + e, N - 1)];
        }
    }
#ifdef TW_STAMP
    PSTAMP0(pst_p1);
    if (lane == 0 && blockIdx.x < 64) { g_stamp3[blockIdx.x][wave][0] = pst_entry; g_stamp3[blockIdx.x][wave][1] = pst_p1; }
#endif
    // ---- records of the block's envs; the image copies must have landed before anyone patches the image
    constexpr bool code_mode = CODE;                            // == (p.flags & TW_F_MATRIX_CODE) != 0: the host picks the instantiation
    uint8_t *img_bytes = reinterpret_cast<uint8_t *>(my_img);
    if (wave < PG && lane < REC) recs[wave * REC + lane] = own_env ? rec_pre : 0;
    if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
    if (blockIdx.x == 0 && tid == 0) *p.abnormal_other = 0;     // stream order: the previous launch's fallback is done
    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): LDS-direct loads retire through the vector-memory counter
    __syncthreads();

#ifdef TW_STAMP
    PSTAMP0(pst_p2);
    if (lane == 0 && blockIdx.x < 64) g_stamp3[blockIdx.x][wave][2] = pst_p2;
#endif
    // ---- wave e verifies env n0+e: scalar regime + planes == closed form (plane bytes: loaded above)
    {
        EnvS s;
        load_env(s, recs + (wave < PG ? wave : 0) * REC);
        bool ok = true;
        if (own_env) {
            ok = pipe_state_ok<V4>(s);
            Dyn d = {s.obx[0], s.pone, s.wall_i1, s.wall_i2, s.patrol, s.o1y[0], s.o2x[0]};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c = lane + 64 * k;
                if (c < NC) ok &= analytic_cell(c - (c / GS) * GS, c / GS, d) == (ty_pre[k] | (co_pre[k] << 8));
            }
        }
        if (__ballot(!ok) != 0ull && lane == 0) atomicOr(p.abnormal, 1);
    }

#ifdef TW_STAMP
    PSTAMP0(pst_p3);
#endif
    // ---- emission constants of this lane.  Record layout (tw_alloc_outputs): the env-step's 2048-byte block is
    // written by exactly two full-wave stores of eight whole 128-byte lines each -- lanes 0..63 matrix floats 0..255,
    // then lanes 0..8 matrix floats 256..291 and lanes 9..63 the 55 image chunks.
    ObsFast of;
    of.rel4[0] = (int)ofw0.x; of.rel4[1] = (int)ofw0.y; of.rel4[2] = (int)ofw0.z; of.rel4[3] = (int)ofw0.w;
    of.rel4[4] = (int)ofw1.x; of.rel4[5] = (int)ofw1.y; of.shift = ofw1.z;
    of.andm[0] = ofw1.w; of.andm[1] = ofw2.x; of.andm[2] = ofw2.y; of.andm[3] = ofw2.z;
    of.orm[0] = ofw2.w; of.orm[1] = ofw3.x; of.orm[2] = ofw3.y; of.orm[3] = ofw3.z;
    of.active = (int)ofw3.w; of.chunk = of.active ? (record ? lane - 9 : lane) : 0;
    // dynamic-cell slot of this lane: 0-2 balls, 3-10 wall blocks, 11-13 patrol column, 14-17 patrol square
    // Every lane decodes the packed record for itself with per-lane shift/mask constants (the scalar unit is
    // shared by the CU's four SIMDs and was the emission waves' bottleneck):
    //   x = xc + ((r >> xsh) & xmask),  y = yc + ((r >> ysh) & ymask),  active = always | ((r >> flagbit) & fmask)
    // Inactive lanes write to a private trash word instead of being masked off (no exec manipulation).
    int dc_xc = 0, dc_yc = 0;
    uint32_t dc_xsh = 0, dc_xmask = 0, dc_ysh = 0, dc_ymask = 0, dc_always = 0, dc_fmask = 0, dc_prebit = 0, dc_postbit = 0;
    uint32_t dc_code = C_BALL, dc_mval = M_BALL;
    const uint8_t dc_mcode = (lane >= 3 && lane < 11) ? 1 : 2;
    if (lane < 3) { dc_always = 1; dc_xc = 6 + lane; dc_xsh = 10; dc_xmask = 3; dc_yc = 8; }                       // balls: x = b0 + k
    else if (lane < 7) { const int k = lane - 3; dc_fmask = 1; dc_prebit = 21; dc_postbit = 22; dc_xc = 4 + (k & 1);
                         dc_yc = 9 + (k >> 1); dc_ysh = 17; dc_ymask = 3; dc_code = C_WALL; dc_mval = M_WALL; }       // block 1: y = i1 + ..
    else if (lane < 11) { const int k = lane - 7; dc_fmask = 1; dc_prebit = 21; dc_postbit = 22; dc_xc = 6 + (k >> 1);
                          dc_xsh = 19; dc_xmask = 3; dc_yc = 11 + (k & 1); dc_code = C_WALL; dc_mval = M_WALL; }      // block 2: x = i2 + ..
    else if (lane < 14) { dc_fmask = 1; dc_prebit = 23; dc_postbit = 24; dc_xc = 12; dc_yc = 3 + (lane - 11);
                          dc_ysh = 12; dc_ymask = 3; }                                                                // patrol column: y = o1y0 + k
    else if (lane < 18) { const int k = lane - 14; dc_fmask = 1; dc_prebit = 23; dc_postbit = 24; dc_xc = 5 + (k & 1);
                          dc_xsh = 14; dc_xmask = 7; dc_yc = 4 + (k >> 1); }                                          // patrol square: x = o2x0 + ..
    // y columns 33, 34 of the x-major image are never inside any window: 66 private trash words, bank-conflict free
    const int trash_g = (lane % GPW) * GPP + 33 + lane / GPW;
    const int mat_lane_off = (MAT_OFF + 4 * lane) * 4;
    const int mat_laneb_off = (MAT_OFF + 4 * (lane < 9 ? 64 + lane : 72)) * 4;

    // ---- LOGIC state of wave 0 (lane e <-> env n0+e): only the fields the closed-form transition touches
    //      (everything else stays in the LDS copy of the record and is written back unchanged).
    // The logic wave is ISSUE-bound (a lone wave gets one instruction of any kind per ~4.3 cycles), so the state is kept
    // in the form that needs the fewest instructions per step -- integers and bit masks, no booleans (every && / || of
    // two lane predicates is an extra scalar instruction on top of the two compares):
    //   posw   agent position, x | y << 5 (the record's low 10 bits; as a shift count the hardware only reads x)
    //   ph2    2 * (step_move % 12): index of the 2-bit tables LG_B0 (ball triple), LG_GL / LG_GH (patrol move gates);
    //          12 = lcm of the 6-step ball cycle and the 4-step gate of the column patrol
    //   c1x2   2 * phase of the column patrol on its 4-step bounce (y, up1): (3,0) (4,0) (5,1) (4,1)
    //   c2x3   3 * phase of the square patrol on its 10-step bounce (x, right2): (5,1) (6,1) .. (9,1) (10,0) (9,0) .. (6,0)
    //          (the patrols never leave these cycles in play; an injected state off them goes to the sequential kernel)
    //   pl2 / ph3   2 / 3 when the column / the square may move this episode (patrol & Update_longitudinal / _horizontal)
    //   npone  ~0 until the wall blocks have dropped, pm ~0 while the patrols exist, w1 / w2 row masks of the dropped wall blocks, dynw the record bits that only
    //          change at a drop / spawn / episode end (pack_record layout: 17-18 i1-9, 19-20 i2-6, 22 pone, 24 patrol,
    //          plus the valid bit), seenw the same flags as the observation of the NEXT step sees them (bits 21, 23)
    const bool lg_active = wave == 0 && lane < PG && n0 + lane < N;
    constexpr uint32_t LG_B0 = 0x019019u;       // (b0 - 6) per phase: 1 2 1 0 0 0 | 1 2 1 0 0 0      (twoarmy_v6.py:96-112)
    constexpr uint32_t LG_GL = 0x3C30F3u;       // column gate open: step_move % 6 in {0, 3} or % 4 == 2 -> phases 0 2 3 6 9 10
    constexpr uint32_t LG_GH = 0xFF3FF3u;       // square gate open: step_move % 6 != 1 -> all phases but 1 and 7
    constexpr uint32_t LG_Y1 = 0x64u;           // column top row - 3 per phase: 0 1 2 1
    constexpr uint32_t LG_X2 = 0x0A72C688u;     // square left column - 5 per phase: 0 1 2 3 4 5 4 3 2 1 (3 bits each)
    constexpr uint32_t GOALW = 14u | (2u << 5);
    struct {
        int step_count, sm_off, risk, max_steps, episodes, last_reward, last_term, last_trunc, i1, i2;
        uint32_t posw, ph2, c1x2, c2x3, pl2, ph3, pm, npone, first_room2, upd_long, upd_horiz, w1, w2, dynw, seenw, t;
    } s;
    bool bad = false;
    {
        const int32_t *r = recs + (lane < PG ? lane : 0) * REC;
        s.posw = (uint32_t)r[TW_AX] | ((uint32_t)r[TW_AY] << 5);
        s.step_count = r[TW_STEP_COUNT]; s.sm_off = r[TW_STEP_MOVE] - s.step_count;
        s.ph2 = 2u * ((uint32_t)r[TW_STEP_MOVE] % 12u);
        s.upd_long = r[TW_UPD_LONG] != 0; s.upd_horiz = r[TW_UPD_HORIZ] != 0;
        s.risk = r[TW_RISK]; s.first_room2 = r[TW_FIRST_ROOM2] != 0; s.max_steps = r[TW_MAX_STEPS];
        s.episodes = r[TW_EPISODES]; s.last_reward = r[TW_LAST_REWARD]; s.last_term = r[TW_LAST_TERM];
        s.last_trunc = r[TW_LAST_TRUNC]; s.t = (uint32_t)r[TW_T];
        const bool pone = r[TW_PONE] != 0;
        s.npone = pone ? 0u : ~0u; s.i1 = r[TW_WALL_I1]; s.i2 = r[TW_WALL_I2];
        const bool patrol = r[TW_PATROL] != 0;
        const int up1 = r[TW_UP1] != 0, right2 = r[TW_RIGHT2] != 0, y1 = r[TW_O1Y], x2 = r[TW_O2X];
        s.w1 = pone ? 0x30u : 0u; s.w2 = pone ? 3u << s.i2 : 0u;
        s.dynw = pone ? (((uint32_t)(s.i1 - 9) & 3u) << 17) | (((uint32_t)(s.i2 - 6) & 3u) << 19) | (1u << 22) : 0u;
        if (patrol) s.dynw |= 1u << 24;
        s.seenw = (s.dynw & ((1u << 22) | (1u << 24))) >> 1;     // record bits 21 / 23: what gen_obs() sees of pone / patrol
        s.dynw |= REC_VALID;                                     // the valid bit rides along
        if (V4 && s.upd_long) s.upd_horiz = 0;    // twoarmy_v4.py:116,148: the first step leaves exactly one of them set
        // patrol phases; without patrols only the coins up1 / right2 are state (phases 0 / 2 and 5 / 0 carry them)
        uint32_t c1 = up1 ? 2u : 0u, c2 = right2 ? 0u : 5u;
        if (V4 && patrol) {
            c1 = up1 ? (y1 == 4 ? 3u : 2u) : (y1 == 4 ? 1u : 0u);
            c2 = right2 ? (uint32_t)(x2 - 5) : (uint32_t)(15 - x2);
            bad |= ((unsigned)(y1 - 3) > 2u) | ((unsigned)(x2 - 5) > 5u) | (up1 ? y1 == 3 : y1 == 5) | (right2 ? x2 == 10 : x2 == 5);
            c1 &= 3u; c2 = c2 > 9u ? 0u : c2;
        }
        s.c1x2 = 2u * c1; s.c2x3 = 3u * c2;
        s.pm = (V4 && patrol) ? ~0u : 0u;
        s.pl2 = (V4 && patrol && s.upd_long) ? 2u : 0u; s.ph3 = (V4 && patrol && s.upd_horiz) ? 3u : 0u;
    }
    const bool policy_idx = (p.flags & TW_F_POLICY_IDX) != 0;

#ifdef TW_STAMP
    PSTAMP0(pst_p4);
#endif
    for (int c0 = 0; c0 < p.T; c0 += PCH) {
        const int len = min(PCH, p.T - c0);
        // the chunk's actions go to LDS up front: the logic wave must never wait on vmcnt (on gfx950 a load
        // retires behind every older store, and the emit waves keep the write queues full)
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = tid + k * 64 * PWAVES;
            if (i >= len * PG) break;
            const int tl = i / PG, e = i - tl * PG;
            const int act_in = (n0 + e < N) ? (c0 == 0 ? act_pre[k] : p.actions[(size_t)(c0 + tl) * N + n0 + e]) : 0;
            ring[i] = (n0 + e < N) ? 0u : REC_VALID;        // padding envs of a ragged last block are "done" from the start
            // The draw counter of an env advances by one per step whatever happens, so every Philox word of
            // the chunk is known up front: all 16 waves compute them in parallel and the serial logic wave
            // only reads one packed word per step (bits 0-1 gate==6 (both), 2-3 wall1, 4-5 wall2, 6-7 spawn, 8 coin A, 9 coin B).
            const uint32_t eid = p.env_id0 + (uint32_t)(n0 + e);
            const uint32_t tnow = (uint32_t)recs[e * REC + TW_T] + (uint32_t)(c0 + tl);
            uint32_t w[4], packed = 0;
            if (V4) {
                draw_block(p.seed_lo, p.seed_hi, eid, tnow, 0, w);
                packed = ((w[TW_S_GATE] % 10u) == 6u ? 3u : 0u) | ((w[TW_S_WALL1] & 3u) << 2) | ((w[TW_S_WALL2] & 3u) << 4) |
                         ((w[TW_S_SPAWN] & 3u) << 6);
            }
            draw_block(p.seed_lo, p.seed_hi, eid, tnow, 1, w);
            packed |= ((w[0] & 1u) << 8) | ((w[1] & 1u) << 9);
            // The action is decoded here as well (Env_transact.env_action 4 -> done, twoarmy_v6.py:85-86 ">= 7 -> left",
            // minigrid.py:1347-1394 move table): the upper half of the word is the signed step of the packed position
            // (dx + 32 dy), bit 10 "the reference raises" (negative action, env actions 4 / 5: AttributeError) -- the
            // serial chain only adds and tests.
            int a = act_in;
            const bool neg = a < 0;
            if (policy_idx && a == 4) a = 6;
            if (a >= 7) a = 0;
            const bool raises = neg || ((0x4Fu >> (a & 7)) & 1u) == 0u;
            const int dx = raises ? 0 : (int)((0x1558u >> (2 * a)) & 3u) - 1, dy = raises ? 0 : (int)((0x1585u >> (2 * a)) & 3u) - 1;
            drw[i] = packed | ((raises ? 1u : 0u) << 10) | ((uint32_t)(dx + 32 * dy) << 16);
        }
        __syncthreads();
#ifdef TW_STAMP
        unsigned long long pst_l = 0, pst_poll = 0, pst_work = 0, pst_tasks = 0, pst_t0 = 0, pst_t1 = 0;
#define PSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#else
#define PSTAMP(v) do { } while (0)
#endif
        if (wave == 0) {
            // ================= LOGIC: the transition in closed form, one step per iteration
            PSTAMP(pst_t0);
            __builtin_amdgcn_s_setprio(3);        // the serial chain must win issue arbitration on its SIMD
            // running LDS pointers (one add per step each); drw has one spare row, so the prefetch of step tl + 1 needs no clamp
            const uint32_t *drp = drw + (lane < PG ? lane : 0);
            uint32_t *rgp = ring + (lane < PG ? lane : 0);
            uint32_t dr_next = *drp, badw = 0u, last_rec = 0u;
            // the record of step t is published at the top of step t + 1 (the LDS write's latency then overlaps that step; a
            // write at the bottom was followed, a few instructions later, by the wait for the next step word, which had to
            // wait for the write as well: LDS operations of a wave retire in order).  Step 0's slot still holds 0: a harmless first write.
            uint32_t pend = 0u, *pendp = rgp;
            // the whole loop sits inside the lane predicate: one exec set-up per chunk instead of an if / else per step
            if (lg_active)
#pragma unroll LG_UNROLL
            for (int tl = 0; tl < len; ++tl) {
                const uint32_t dr = dr_next;
                drp += PG;
                dr_next = *drp;                                           // next step's word: its LDS latency leaves the chain
                badw |= dr;                                               // bit 10: the reference raises on this action
                // step_move += 1 (twoarmy_v6.py:88); row-8 balls (:96-112): the triple's x is a pure function of
                // step_move % 6 (7 at reset, +1 for {0,1}, -1 for {2,3}, 0 for {4,5}); verified against the record at launch
                s.ph2 = s.ph2 == 22u ? 0u : s.ph2 + 2u;
                const uint32_t b0o = (LG_B0 >> s.ph2) & 3u;
                const uint32_t balls = 0x1C0u << b0o;                     // 7 << b0
                if (V4) {                                                 // twoarmy_v4.py:115-176
                    // the column moves when step_move % 6 in {0, 3} (or % 4 == 2, or the drawn gate), the square when
                    // step_move % 6 != 1 (or the gate) -- two bit tables over the phase, OR-ed with the gate draw (bits 0-1
                    // of the step word), AND-ed with "this patrol may move": the phase increment itself
                    s.c1x2 = (s.c1x2 + (((LG_GL >> s.ph2) | dr) & s.pl2)) & 6u;
                    s.c2x3 += ((LG_GH >> s.ph2) | dr) & s.ph3;
                    s.c2x3 = s.c2x3 == 30u ? 0u : s.c2x3;
                }
                uint32_t y1o = (LG_Y1 >> s.c1x2) & 3u, x2o = (LG_X2 >> s.c2x3) & 7u;   // column top row - 3, square left column - 5
                // MiniGridEnv.step (minigrid.py:1333-1441): is the target cell free?  One bitmask per grid row (bit x set
                // = cell (x, ty) blocks): border / row-8 walls + ball triple + dropped 2x2 wall blocks (+ patrol column and
                // square).  The goal cell (14, 2) is never blocked.
                s.step_count += 1;
                const uint32_t tposw = s.posw + (uint32_t)((int32_t)dr >> 16);
                *pendp = pend;                                            // previous step's record (see above)
                const int ty = (int)(tposw >> 5);
                uint32_t rb = ty == 8 ? (0x1F83Fu | balls) : 0x10001u;
                rb = (unsigned)(ty - 1) > 14u ? 0x1FFFFu : rb;
                rb |= (unsigned)(ty - s.i1) <= 1u ? s.w1 : 0u;
                rb |= (unsigned)(ty - 11) <= 1u ? s.w2 : 0u;
                if (V4) {
                    rb |= (unsigned)(ty - 3 - (int)y1o) <= 2u ? (s.pm & 0x1000u) : 0u;
                    rb |= (unsigned)(ty - 4) <= 1u ? (s.pm & (0x60u << x2o)) : 0u;
                }
                s.posw = __builtin_amdgcn_ubfe(rb, tposw, 1u) ? s.posw : tposw;    // bit tx of the row mask (the hardware reads 5 bits of the offset)
                const uint32_t terminated = tposw == GOALW;
                uint32_t truncated = s.step_count >= s.max_steps;
                const uint32_t seen = s.seenw;                            // what gen_obs() saw: pone / patrol before this step's drop / spawn
                const int ay = (int)(s.posw >> 5);
                // ---- after gen_obs(): wall drop, patrol spawn (twoarmy_v6.py:182-198, v4:181-225); one test for both:
                // "still armed" = the sign bit of ~pone-mask, or patrols absent (v4) and ay <= 8
                const uint32_t armed = V4 ? (s.npone | ((s.posw - (9u << 5)) & ~s.pm)) : s.npone;
                if (LG_RARE((int32_t)armed < 0)) {
                    if (s.npone && ((s.posw & 31u) > 3u || ay < 14)) {
                        s.i1 = V4 ? 9 + (int)((dr >> 2) & 3u) : 11;
                        s.i2 = V4 ? 6 + (int)((dr >> 4) & 3u) : 8;
                        s.npone = 0u;
                        s.w1 = 0x30u; s.w2 = 3u << s.i2;
                        s.dynw = (s.dynw & ~(15u << 17)) | ((uint32_t)(s.i1 - 9) << 17) | ((uint32_t)(s.i2 - 6) << 19) | (1u << 22);
                        s.seenw |= 1u << 21;
                    }
                    if (V4 && s.pm == 0u && ay <= 8) {                    // column at rows 4..6 keeping up1, square at x = 6 + draw keeping right2
                        const uint32_t dq = (dr >> 6) & 3u;
                        s.c1x2 = 2u + (s.c1x2 & 4u);
                        s.c2x3 = s.c2x3 < 15u ? 3u + 3u * dq : 27u - 3u * dq;
                        y1o = 1u; x2o = 1u + dq;
                        s.pm = ~0u; s.pl2 = s.upd_long ? 2u : 0u; s.ph3 = s.upd_horiz ? 3u : 0u;
                        s.dynw |= 1u << 24;
                        s.seenw |= 1u << 23;
                    }
                }
                // shaped reward (twoarmy_v6.py:231-294) from two masks of the agent's row: bit x set = standing on x is a
                // collision (a row-8 ball; v4: a patrol ball, twoarmy_v4.py:242-280) / is "at risk" (under a row-8 ball;
                // v4: left of the column, left / right of or under the square).  A collision outranks a risk.
                uint32_t hitm = ay == 8 ? balls : 0u, riskm = ay == 9 ? balls : 0u;
                if (V4) {
                    const uint32_t colr = (unsigned)(ay - 3 - (int)y1o) <= 2u ? s.pm : 0u, sqr = (unsigned)(ay - 4) <= 1u ? s.pm : 0u;
                    const uint32_t sq = 0x60u << x2o;
                    hitm |= (colr & 0x1000u) | (sqr & sq);
                    riskm |= (colr & 0x800u) | (sqr & ((sq << 1) | (sq >> 1))) | (ay == 6 ? s.pm & sq : 0u);
                }
                const uint32_t hit = __builtin_amdgcn_ubfe(hitm, s.posw, 1u), risky = __builtin_amdgcn_ubfe(riskm, s.posw, 1u);
                uint32_t reward = hit ? (uint32_t)R_HIT : risky;          // R_RISK == 1, R_STEP == 0
                truncated |= hit;
                const uint32_t room2 = ay == 7 ? s.first_room2 : 0u;
                reward = room2 ? (uint32_t)R_ROOM2 : reward;
                s.first_room2 &= ~room2;
                const uint32_t isrisk = reward == (uint32_t)R_RISK;
                s.risk += (int)isrisk;
                truncated |= isrisk & (uint32_t)(s.risk > 5);
                // record of this env-step (pack_record's layout): a chain of shift-ORs onto the registers that already hold
                // their fields in place (reward | terminated | truncated are bits 25-29: one 5-bit field); single-word
                // publish, the logic wave issues NO vector-memory op in its loop
                const uint32_t rtt = lshl_or<4>(truncated, terminated ? ((uint32_t)R_GOAL | 8u) : reward);
                uint32_t rec = lshl_or<10>(b0o, s.posw | s.dynw | seen);
                if (V4) rec = lshl_or<14>(x2o, lshl_or<12>(y1o, rec));
                rec = lshl_or<25>(rtt, rec);
                if (LG_RARE((terminated | truncated) != 0u)) {            // twoarmy_v6.py:296-318 + auto-reset
                    const uint32_t ca = (dr >> 8) & 1u, cb = (dr >> 9) & 1u;
                    s.ph2 = 0u; s.sm_off = 0; s.first_room2 = 1u; s.risk = 0;
                    s.c1x2 = 4u - 4u * ca; s.c2x3 = 15u - 15u * ca;      // up1 = 1 - ca, right2 = ca; no patrol until the next spawn
                    s.upd_horiz = 1u - cb; s.upd_long = cb;
                    s.episodes += 1;
                    s.npone = ~0u; s.pm = 0u; s.pl2 = 0u; s.ph3 = 0u; s.w1 = 0u; s.w2 = 0u; s.dynw = REC_VALID; s.seenw = 0u;
                    s.posw = 3u | (15u << 5); s.step_count = 0;
                }
                pend = rec; pendp = rgp;
                rgp += PG;
            }
            if (lg_active) { *pendp = pend; last_rec = pend; }
            if (lg_active) {                                              // what the loop left for the end of the chunk
                s.t += (uint32_t)len;                                     // one draw-counter tick per step, whatever happens
                s.last_reward = (int)((last_rec >> 25) & 7u); s.last_term = (int)((last_rec >> 28) & 1u);
                s.last_trunc = (int)((last_rec >> 29) & 1u);
                bad |= ((badw >> 10) & 1u) != 0u;
            }
            if (__ballot(bad) != 0ull && lane == 0) atomicOr(p.abnormal, 1);
            __builtin_amdgcn_s_setprio(0);
            PSTAMP(pst_t1);
#ifdef TW_STAMP
            pst_l = pst_t1 - pst_t0;
            if (c0 == 0) pst_lstart = pst_t0;
            pst_lend = pst_t1;
            if (lane == 0 && blockIdx.x < 64) g_stamp[blockIdx.x][0] = pst_l;
#endif
        }
        // ================= EMIT: pull groups of PGRP (step, env) tasks
        const int ngroups = len * (PG / PGRP);
        while (true) {
            int kg = 0;
            if (lane == 0) kg = atomicAdd(const_cast<int *>(&ctrl[1]), 1);
            kg = __builtin_amdgcn_readfirstlane(kg);
            if (kg >= ngroups) break;
            const int tl = kg / (PG / PGRP), e0 = (kg - tl * (PG / PGRP)) * PGRP;
            // lanes 0..PGRP-1 poll their record until the logic wave has published it (valid bit in the same word);
            // the wave that draws a step's first group polls all PG records and writes the step's scalar outputs
            const int npoll = e0 == 0 ? PG : PGRP;
            uint32_t rv = REC_VALID;
            PSTAMP(pst_t0);
            while (true) {
                if (lane < npoll) rv = const_cast<volatile uint32_t *>(ring)[tl * PG + e0 + lane];
                if (__ballot((rv & REC_VALID) == 0u) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
            PSTAMP(pst_t1);
#ifdef TW_STAMP
            pst_poll += pst_t1 - pst_t0; pst_tasks += PGRP;
#endif
            if (e0 == 0 && lane < PG && n0 + lane < N) {   // scalar outputs of the step's PG env-steps: one lane each, contiguous
                const size_t srow = (size_t)(c0 + tl) * N + n0 + lane;
                p.reward[srow] = reward_value((int)((rv >> 25) & 7u));
                p.term[srow] = (uint8_t)((rv >> 28) & 1u);
                p.trunc[srow] = (uint8_t)((rv >> 29) & 1u);
                reinterpret_cast<float2 *>(p.pos)[srow] = make_float2((float)((rv >> 5) & 31u), (float)(rv & 31u));
            }
            const size_t grow = (size_t)(c0 + tl) * N + n0 + e0;          // first output row of the group
            uint8_t *obs_g = p.obs + grow * (size_t)p.obs_pitch;
            float *mat_g = p.matrix + grow * (size_t)p.mat_pitch;
#pragma unroll 1
            for (int j = 0; j < PGRP; ++j) {
                if (n0 + e0 + j >= N) break;
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rv, j);
                const int ax = r & 31, ay = (r >> 5) & 31;
                uint8_t *obs_dst = obs_g + (size_t)j * p.obs_pitch;
                float *mat_dst = mat_g + (size_t)j * p.mat_pitch;
                uint8_t *matc_dst = reinterpret_cast<uint8_t *>(p.matrix) + (grow + j) * (size_t)p.mat_pitch;
#ifdef TW_PIPE_NO_EMIT
                if (ax == 99) p.obs[0] = (uint8_t)(ax + ay);
                continue;
#endif
                // this lane's dynamic cell, decoded from the record with per-lane shift/mask constants
                const int x = dc_xc + (int)((r >> dc_xsh) & dc_xmask);
                const int y = dc_yc + (int)((r >> dc_ysh) & dc_ymask);
                const bool pre = (dc_always | ((r >> dc_prebit) & dc_fmask)) != 0u;
                const int gcell = gpi(x, y), mcell = MAT_OFF + y * GS + x;
                const int gidx = pre ? gcell : trash_g, midx = pre ? mcell : trash_g;
                const int ca = MAT_OFF + ay * GS + ax;
                const int bidx = pre ? MATC_OFF * 4 + y * GS + x : trash_g * 4;       // byte image of the matrix codes
                const int bca = lane == 0 ? MATC_OFF * 4 + ay * GS + ax : trash_g * 4;
                my_img[gidx] = dc_code;                                  // unconditional: idle lanes hit their trash word
                // code frames never touch the float matrix image, float frames never the code bytes (the emission is
                // issue-bound with code frames: every LDS operation saved counts)
                if (code_mode) img_bytes[bidx] = dc_mcode; else my_img[midx] = dc_mval;
                wave_sync();
                if ((((r >> 21) ^ (r >> 22)) & 5u) == 0u) {
                    // common case: obs and matrix see the same grid -> one batch of gathers, then pack + store
                    if (code_mode) img_bytes[bca] = 3;                    // after the dynamic cells: the agent wins (0.3 / code 3)
                    else my_img[lane == 0 ? ca : trash_g] = M_AGENT;
                    wave_sync();
                    const char *base = reinterpret_cast<const char *>(my_img + gpi(ax, ay));
                    uint32_t c[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) c[k] = *reinterpret_cast<const uint32_t *>(base + of.rel4[k]);
                    uint4 m0 = make_uint4(0u, 0u, 0u, 0u), m1 = m0;
                    if (code_mode) {
                        m0 = *reinterpret_cast<const uint4 *>(img_bytes + MATC_OFF * 4 + 16 * (lane < MATC_BYTES / 16 ? lane : 0));
                    } else {
                        m0 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(my_img) + mat_lane_off);
                        m1 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(my_img) + mat_laneb_off);
                    }
                    const uint32_t w0 = c[0] | (c[1] << 24), w1 = (c[1] >> 8) | (c[2] << 16), w2 = (c[2] >> 16) | (c[3] << 8);
                    const uint32_t w3 = c[4] | (c[5] << 24), w4 = c[5] >> 8;
                    uint4 o;
                    o.x = (__builtin_amdgcn_alignbit(w1, w0, of.shift) & of.andm[0]) | of.orm[0];
                    o.y = (__builtin_amdgcn_alignbit(w2, w1, of.shift) & of.andm[1]) | of.orm[1];
                    o.z = (__builtin_amdgcn_alignbit(w3, w2, of.shift) & of.andm[2]) | of.orm[2];
                    o.w = (__builtin_amdgcn_alignbit(w4, w3, of.shift) & of.andm[3]) | of.orm[3];
                    if (record) {
                        uint8_t *rec_dst = reinterpret_cast<uint8_t *>(mat_dst);
                        store16(rec_dst + 16 * lane, m0);
                        uint4 b;
                        b.x = lane < 9 ? m1.x : o.x; b.y = lane < 9 ? m1.y : o.y;
                        b.z = lane < 9 ? m1.z : o.z; b.w = lane < 9 ? m1.w : o.w;
                        if ((lane < 9) | of.active) store16(rec_dst + 1024 + 16 * lane, b);
                    } else if (!code_mode) {
                        if (of.active) store16(obs_dst + 16 * lane, o);
                        store16(mat_dst + 4 * lane, m0);
                        if (lane < 9) store16(mat_dst + 4 * (64 + lane), m1);
                    } else {
                        if (of.active) store16(obs_dst + 16 * lane, o);
                        if (lane < MATC_BYTES / 16) store16(matc_dst + 16 * lane, m0);
                    }
                    wave_sync();
                    my_img[gidx] = C_EMPTY;                               // un-patch (trash words may hold anything)
                    if (code_mode) img_bytes[bidx] = 0; else my_img[midx] = M_FREE;
                    wave_sync();
                    if (code_mode) img_bytes[bca] = 0;                    // the agent only ever stands on free / goal / ball cells
                    else my_img[lane == 0 ? ca : trash_g] = M_FREE;
                } else {
                    // wall drop / patrol spawn happened in this very step: the matrix sees it, the observation did not
                    const bool post = (dc_always | ((r >> dc_postbit) & dc_fmask)) != 0u;
                    emit_obs_fast(my_img, ax, ay, of, lane, obs_dst);
                    wave_sync();
                    if (post) { my_img[gcell] = dc_code; my_img[mcell] = dc_mval; mcb(my_img)[y * GS + x] = dc_mcode; }
                    wave_sync();
                    if (code_mode) emit_codes(my_img, ax, ay, lane, matc_dst, true);
                    else emit_matrix_fast(my_img, ax, ay, lane, mat_dst);
                    wave_sync();
                    if (post | pre) { my_img[gcell] = C_EMPTY; my_img[mcell] = M_FREE; mcb(my_img)[y * GS + x] = 0; }
                }
                wave_sync();
            }
#ifdef TW_STAMP
            PSTAMP(pst_t0);
            pst_work += pst_t0 - pst_t1;
#endif
        }
#ifdef TW_STAMP
        if (wave == 1 && lane == 0 && blockIdx.x < 64) {
            g_stamp[blockIdx.x][1] = pst_poll; g_stamp[blockIdx.x][2] = pst_work; g_stamp[blockIdx.x][3] = pst_tasks;
        }
        if (lane == 0 && blockIdx.x < 64) {
            g_stamp2[blockIdx.x][wave][0] = pst_tasks; g_stamp2[blockIdx.x][wave][1] = pst_poll; g_stamp2[blockIdx.x][wave][2] = pst_work;
        }
#endif
        __syncthreads();
        if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
        __syncthreads();
    }

    // ---- write the final state (records from the logic lanes, planes from the closed form)
    if (wave == 0 && lane < PG) {
        int32_t *r = recs + lane * REC;
        const int c1 = (int)(s.c1x2 >> 1), c2 = (int)(s.c2x3 / 3u);
        const int up1 = c1 >> 1, right2 = c2 < 5;           // the episode-end coins live in the patrol phases (v6 too)
        const int patrol = s.pm != 0u, o1y0 = 3 + (int)((LG_Y1 >> s.c1x2) & 3u), o2x0 = 5 + (int)((LG_X2 >> s.c2x3) & 7u);
        r[TW_AX] = (int)(s.posw & 31u); r[TW_AY] = (int)(s.posw >> 5); r[TW_STEP_COUNT] = s.step_count;
        r[TW_STEP_MOVE] = s.step_count + s.sm_off;
        r[TW_UP1] = up1; r[TW_RIGHT2] = right2; r[TW_UPD_LONG] = (int)s.upd_long; r[TW_UPD_HORIZ] = (int)s.upd_horiz;
        r[TW_RISK] = s.risk; r[TW_FIRST_ROOM2] = (int)s.first_room2; r[TW_EPISODES] = s.episodes;
        r[TW_LAST_REWARD] = s.last_reward; r[TW_LAST_TERM] = s.last_term; r[TW_LAST_TRUNC] = s.last_trunc;
        r[TW_T] = (int32_t)s.t; r[TW_ERROR] = 0;
        r[TW_PONE] = s.npone == 0u; r[TW_PATROL] = patrol; r[TW_WALL_I1] = s.i1; r[TW_WALL_I2] = s.i2;
        const int b0_end = 6 + (int)((LG_B0 >> s.ph2) & 3u);               // ball triple as a function of step_move % 6
#pragma unroll
        for (int k = 0; k < 3; ++k) { r[TW_OBX + k] = b0_end + k; r[TW_OBY + k] = 8; }
        if (V4) {
            r[TW_O1_VALID] = patrol; r[TW_O2_VALID] = patrol;
            if (patrol) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { r[TW_O1X + k] = 12; r[TW_O1Y + k] = o1y0 + k; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { r[TW_O2X + k] = o2x0 + (k & 1); r[TW_O2Y + k] = 4 + (k >> 1); }
            }
        }
    }
    __syncthreads();
    if (wave < PG && n0 + wave < N) {
        const int32_t *r = recs + wave * REC;
        if (lane < REC) p.rec_out[(size_t)(n0 + wave) * REC + lane] = r[lane];
        Dyn fd = {r[TW_OBX], r[TW_PONE], r[TW_WALL_I1], r[TW_WALL_I2], r[TW_PATROL], r[TW_O1Y], r[TW_O2X]};
        const size_t gb = (size_t)(n0 + wave) * NC;
        for (int c = lane; c < NC; c += 64) {
            const uint32_t v = analytic_cell(c - (c / GS) * GS, c / GS, fd);
            p.type_out[gb + c] = (uint8_t)v;
            p.colour_out[gb + c] = (uint8_t)(v >> 8);
        }
    }
#ifdef TW_STAMP
    if (wave == 0) {
        unsigned long long pst_exit;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_exit) :: "memory");
        if (lane == 0 && blockIdx.x < 64) {
            g_stamp[blockIdx.x][4] = pst_lstart - pst_entry;       // prologue: images, verification, staging
            g_stamp[blockIdx.x][5] = pst_exit - pst_lend;          // from the logic wave's last record to the kernel's end
            g_stamp[blockIdx.x][6] = pst_exit - pst_entry;         // whole kernel as wave 0 sees it
            g_stamp2[blockIdx.x][0][0] = pst_p1 - pst_entry; g_stamp2[blockIdx.x][0][1] = pst_p2 - pst_p1;
            g_stamp2[blockIdx.x][0][2] = pst_p3 - pst_p2; g_stamp2[blockIdx.x][1][0] = pst_p4 - pst_p3; g_stamp2[blockIdx.x][1][1] = pst_lstart - pst_p4;
        }
    }
#endif
}

// ---------------------------------------------------------------- init / reset / obs-only kernels (wave per env)
// mode 0: Twoarmy __init__ (flags armed, twoarmy_v6.py:15-25) + reset;  mode 1: MiniGridEnv.reset only.
__global__ __launch_bounds__(64) void tw_reset_kernel(Params p, const uint8_t *mask, int mode) {
    __shared__ uint32_t cells[NC + 3];
    __shared__ uint32_t stage[STAGE_WORDS];
    const int n = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[n]) return;
    int32_t *r = p.rec + (size_t)n * REC;
    if (lane == 0) {
        if (mode == 0) {
            for (int k = 0; k < REC; ++k) r[k] = 0;
            r[TW_MAX_STEPS] = 50;
            r[TW_UPD_LONG] = 1; r[TW_RIGHT2] = 1; r[TW_FIRST_ROOM2] = 1;
        }
        for (int k = 0; k < 3; ++k) { r[TW_OBX + k] = 7 + k; r[TW_OBY + k] = 8; }
        r[TW_O1_VALID] = 0; r[TW_O2_VALID] = 0;
        r[TW_AX] = 3; r[TW_AY] = 15; r[TW_DIR] = 3; r[TW_GOAL_X] = 14; r[TW_GOAL_Y] = 2;
        r[TW_STEP_COUNT] = 0; r[TW_ERROR] = 0;
    }
    for (int c = lane; c < NC; c += 64) {
        const int y = c / GS, x = c - y * GS;
        const uint32_t v = gen_cell(x, y);
        cells[c] = v;
        p.type[(size_t)n * NC + c] = (uint8_t)v;
        p.colour[(size_t)n * NC + c] = (uint8_t)(v >> 8);
    }
    wave_sync();
    const uint32_t *cc = cells;
    if (p.obs)
        emit_obs_generic([cc](int x, int y) { return cc[y * GS + x]; }, stage, lane, p.view, 3, 15, 3,
                         p.obs + (size_t)n * p.obs_pitch);
}

__global__ __launch_bounds__(64) void tw_gen_obs_kernel(Params p) {
    __shared__ uint32_t cells[NC + 3];
    __shared__ uint32_t stage[STAGE_WORDS];
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int c = lane; c < NC; c += 64)
        cells[c] = (uint32_t)p.type[(size_t)n * NC + c] | ((uint32_t)p.colour[(size_t)n * NC + c] << 8);
    wave_sync();
    const int32_t *r = p.rec + (size_t)n * REC;
    const int ax = __builtin_amdgcn_readfirstlane(r[TW_AX]), ay = __builtin_amdgcn_readfirstlane(r[TW_AY]);
    const int dir = __builtin_amdgcn_readfirstlane(r[TW_DIR]);
    const uint32_t *cc = cells;
    emit_obs_generic([cc](int x, int y) { return cc[y * GS + x]; }, stage, lane, p.view, ax, ay, dir,
                     p.obs + (size_t)n * p.obs_pitch);
}

__global__ void tw_fill_actions_kernel(Params p, int32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = p.T * p.n_envs;
    if (i >= total) return;
    const int tt = i / p.n_envs, n = i - tt * p.n_envs;
    const uint32_t t = (uint32_t)p.rec[(size_t)n * REC + TW_T] + (uint32_t)tt;
    uint32_t w[4];
    draw_block(p.seed_lo, p.seed_hi, p.env_id0 + (uint32_t)n, t, 2, w);
    out[i] = (int32_t)(w[0] % 5u);
}

int g_last_hip_error = 0;
char g_last_error_msg[256] = "";



}  // namespace

// ======================================================================= host side / C ABI
struct tw_engine {
    int variant, n_envs, view, device;
    uint64_t seed;
    uint32_t env_id0;
    uint8_t *type, *colour;         // current state (SoA planes + records)
    int32_t *rec;
    uint8_t *type2, *colour2;       // ping-pong partner written by a pipelined launch, swapped in afterwards
    int32_t *rec2;
    int *abnormal;                  // two device flags of the pipelined kernel, used alternately
    int parity;
    int envs_per_wave;              // 0 = auto
    int pipeline;                   // 1 = use the pipelined kernel when eligible (TW_PIPELINE=0 disables)
    int *fb_count;                  // device counter: pipelined launches re-run by the sequential fallback
    int slab_backing;               // how tw_alloc_outputs backs its slab (0 hipMalloc, 1 mapped 2 MiB granules, ...)
    uint32_t *pipe_tab;             // constant tables of the pipelined kernel (static image, per-lane emission constants)
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int hip_fail_at(hipError_t e, const char *what, int line) {
    g_last_hip_error = (int)e;
    snprintf(g_last_error_msg, sizeof(g_last_error_msg), "%s (line %d): %s", what, line, hipGetErrorString(e));
    return TW_E_HIP;
}
#define hip_fail(e) hip_fail_at((e), "hip", __LINE__)
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_fail_at(_e, #expr, __LINE__); } while (0)

Params base_params(const tw_engine *e) {
    Params p;
    memset(&p, 0, sizeof(p));
    p.type = e->type; p.colour = e->colour; p.rec = e->rec;
    p.type_out = e->type; p.colour_out = e->colour; p.rec_out = e->rec;
    p.abnormal = e->abnormal; p.abnormal_other = e->abnormal + 1; p.only_if_flagged = 0;
    p.fb_count = e->fb_count;
    p.pipe_tab = e->pipe_tab;
    p.n_envs = e->n_envs; p.view = e->view; p.variant = e->variant;
    p.seed_lo = (uint32_t)e->seed; p.seed_hi = (uint32_t)(e->seed >> 32);
    p.env_id0 = e->env_id0;
    p.T = 1;
    p.obs_pitch = e->view * e->view * 3;
    p.mat_pitch = NC;
    return p;
}

bool view_ok(int v) { return v >= 3 && v <= GS && (v & 1); }

// envs per wavefront: amortise the lane-per-env logic while keeping >= ~2 waves per SIMD (1024 SIMDs)
int pick_envs_per_wave(const tw_engine *e) {
    if (e->envs_per_wave > 0) return e->envs_per_wave;
    if (e->n_envs >= 2048) return 2;       // measured best at 4096 envs on MI355X
    return 1;
}

bool params_fast(const tw_engine *e, const Params &p, bool allow_codes) {
    const int chunk = ((e->view * e->view * 3 + 15) >> 4) << 4;
    const bool codes = (p.flags & TW_F_MATRIX_CODE) != 0;
    if (codes && !allow_codes) return false;
    const bool mat_ok = codes ? ((p.mat_pitch & 15) == 0 && p.mat_pitch >= MATC_BYTES)
                              : ((p.mat_pitch & 3) == 0 && p.mat_pitch >= MAT_WORDS);
    return p.obs && p.matrix && p.pos && p.reward && p.term && p.trunc && p.actions && !p.draws &&
           ((uintptr_t)p.obs & 15u) == 0 && (p.obs_pitch & 15) == 0 && p.obs_pitch >= chunk &&
           ((uintptr_t)p.matrix & 15u) == 0 && mat_ok;
}

template <int E>
void launch_variant(const tw_engine *e, const Params &p, hipStream_t st) {
    int grid = (e->n_envs + E - 1) / E;
    if (p.only_if_flagged && grid > 256) grid = 256;        // grid-stride fallback launch (see tw_rollout_kernel)
    const bool fast = params_fast(e, p, false);
    if (e->variant == 4) {
        if (fast) hipLaunchKernelGGL((tw_rollout_kernel<E, 4, true>), dim3(grid), dim3(64), 0, st, p);
        else hipLaunchKernelGGL((tw_rollout_kernel<E, 4, false>), dim3(grid), dim3(64), 0, st, p);
    } else {
        if (fast) hipLaunchKernelGGL((tw_rollout_kernel<E, 6, true>), dim3(grid), dim3(64), 0, st, p);
        else hipLaunchKernelGGL((tw_rollout_kernel<E, 6, false>), dim3(grid), dim3(64), 0, st, p);
    }
}

constexpr int PIPE_MIN_T = 8;
constexpr size_t PIPE_LDS_BYTES = (size_t)(PWAVES * ENV_WORDS + PCH * PG_MAX + PG_MAX * REC + 4 + (PCH + 1) * PG_MAX) * 4;   // ~117 KB; drw has a spare row

int launch_sequential(const tw_engine *e, const Params &p, hipStream_t st) {
    switch (pick_envs_per_wave(e)) {
    case 1: launch_variant<1>(e, p, st); break;
    case 2: launch_variant<2>(e, p, st); break;
    case 4: launch_variant<4>(e, p, st); break;
    default: return TW_E_ARG;
    }
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

int launch_rollout(tw_engine *e, int T, const int32_t *actions, const uint32_t *draws, uint8_t *obs, int obs_pitch,
                   float *matrix, int mat_pitch, float *pos, float *reward, uint8_t *term, uint8_t *trunc, int flags,
                   hipStream_t st) {
    Params p = base_params(e);
    p.T = T; p.actions = actions; p.draws = draws; p.obs = obs; p.matrix = matrix; p.pos = pos;
    p.reward = reward; p.term = term; p.trunc = trunc; p.flags = flags;
    if (obs_pitch > 0) p.obs_pitch = obs_pitch;
    if (mat_pitch > 0) p.mat_pitch = mat_pitch;
    if (p.obs_pitch < e->view * e->view * 3 || p.mat_pitch < NC) return TW_E_ARG;   // pitch: floats, or bytes with TW_F_MATRIX_CODE
    p.record = !(flags & TW_F_MATRIX_CODE) && obs && matrix && p.obs_pitch == p.mat_pitch * 4 &&
               obs == reinterpret_cast<uint8_t *>(matrix) + REC_OBS_OFF && ((uintptr_t)matrix & 15u) == 0 &&
               p.obs_pitch >= REC_OBS_OFF + ((((e->view * e->view * 3) + 15) >> 4) << 4);
    const bool pipe = e->pipeline && T >= PIPE_MIN_T && (flags & TW_F_AUTORESET) && params_fast(e, p, true);
    if (!pipe) return launch_sequential(e, p, st);
    // pipelined launch: cur -> next, with the sequential kernel as a flag-gated fallback from the same input
    p.type_out = e->type2; p.colour_out = e->colour2; p.rec_out = e->rec2;
    p.abnormal = e->abnormal + e->parity;
    p.abnormal_other = e->abnormal + (e->parity ^ 1);
    e->parity ^= 1;
    // envs per workgroup: 16 from 4096 envs up; fewer for small batches so that ~256 workgroups exist (one per CU)
    int pg = 2;
    while (pg < PG_MAX && (e->n_envs + pg - 1) / pg > PIPE_WG_TARGET) pg <<= 1;
    // One round of 16-env workgroups keeps the whole chip in lock step: every CU in its prologue, then every CU storing,
    // then every CU in its tail.  Two rounds of 8-env workgroups desynchronise the CUs after the first round (measured
    // at 4096 envs, v6: 3-4 % faster on two boxes, v4: equal; from 8192 envs up the 16-env groups already make several
    // rounds, below ~2000 envs a second round would double the logic chain, which bounds the launch there).
    if (pg == 16 && (e->n_envs + 15) / 16 <= PIPE_WG_TARGET) pg = 8;
    const int grid = (e->n_envs + pg - 1) / pg;
#define TW_PIPE_LAUNCH(VAR, PGV, LY) \
    hipLaunchKernelGGL((tw_pipe_kernel<VAR, PGV, LY>), dim3(grid), dim3(64 * PWAVES), PIPE_LDS_BYTES, st, p)
#define TW_PIPE_LAUNCH_PG(VAR, LY) do { \
        if (pg == 16) TW_PIPE_LAUNCH(VAR, 16, LY); else if (pg == 8) TW_PIPE_LAUNCH(VAR, 8, LY); \
        else if (pg == 4) TW_PIPE_LAUNCH(VAR, 4, LY); else TW_PIPE_LAUNCH(VAR, 2, LY); } while (0)
    const int layout = (flags & TW_F_MATRIX_CODE) ? 2 : (p.record ? 1 : 0);
    if (e->variant == 4) {
        if (layout == 2) TW_PIPE_LAUNCH_PG(4, 2); else if (layout == 1) TW_PIPE_LAUNCH_PG(4, 1); else TW_PIPE_LAUNCH_PG(4, 0);
    } else {
        if (layout == 2) TW_PIPE_LAUNCH_PG(6, 2); else if (layout == 1) TW_PIPE_LAUNCH_PG(6, 1); else TW_PIPE_LAUNCH_PG(6, 0);
    }
#undef TW_PIPE_LAUNCH_PG
#undef TW_PIPE_LAUNCH
    HIP_TRY(hipGetLastError());
    p.only_if_flagged = 1;
#ifdef TW_DEBUG_NO_FALLBACK_LAUNCH
    int rc = TW_OK;                              // diagnostic build only: what the flag-gated launch costs
#else
    int rc = launch_sequential(e, p, st);
#endif
    if (rc != TW_OK) return rc;
    uint8_t *t8; int32_t *t32;
    t8 = e->type; e->type = e->type2; e->type2 = t8;
    t8 = e->colour; e->colour = e->colour2; e->colour2 = t8;
    t32 = e->rec; e->rec = e->rec2; e->rec2 = t32;
    return TW_OK;
}

}  // namespace

extern "C" {

int tw_create(tw_engine **out, int variant, int n_envs, int view_size, int device_id, uint64_t seed,
              uint32_t env_id0) {
    if (!out || (variant != 4 && variant != 6) || n_envs <= 0 || !view_ok(view_size)) return TW_E_ARG;
    DeviceGuard g(device_id);
    if (!g.ok) return hip_fail(hipErrorInvalidDevice);
    tw_engine *e = (tw_engine *)calloc(1, sizeof(tw_engine));
    if (!e) return TW_E_NOMEM;
    e->variant = variant; e->n_envs = n_envs; e->view = view_size; e->device = device_id;
    e->seed = seed; e->env_id0 = env_id0;
    const char *epw = getenv("TW_ENVS_PER_WAVE");
    e->envs_per_wave = epw ? atoi(epw) : 0;
    const char *pl = getenv("TW_PIPELINE");
    e->pipeline = pl ? atoi(pl) : 1;
    e->slab_backing = 1;            // 2 MiB chunks created one by one, mapped in creation order (see slab_alloc)
    hipError_t rr[8];
    rr[0] = hipMalloc((void **)&e->type, (size_t)n_envs * NC);
    rr[1] = hipMalloc((void **)&e->colour, (size_t)n_envs * NC);
    rr[2] = hipMalloc((void **)&e->rec, (size_t)n_envs * REC * sizeof(int32_t));
    rr[3] = hipMalloc((void **)&e->type2, (size_t)n_envs * NC);
    rr[4] = hipMalloc((void **)&e->colour2, (size_t)n_envs * NC);
    rr[5] = hipMalloc((void **)&e->rec2, (size_t)n_envs * REC * sizeof(int32_t));
    rr[6] = hipMalloc((void **)&e->abnormal, 3 * sizeof(int));
    rr[7] = hipMalloc((void **)&e->pipe_tab, (size_t)PT_WORDS * sizeof(uint32_t));
    for (int i = 0; i < 8; ++i)
        if (rr[i] != hipSuccess) { hipError_t bad = rr[i]; tw_destroy(e); return hip_fail(bad); }
    {
        hipError_t me = hipMemset(e->abnormal, 0, 3 * sizeof(int));
        e->fb_count = e->abnormal + 2;
        if (me != hipSuccess) { tw_destroy(e); return hip_fail(me); }
    }
    {   // the pipelined kernel needs > 64 KB of dynamic LDS
#define TW_PIPE_K(VAR, LY) reinterpret_cast<const void *>(&tw_pipe_kernel<VAR, 16, LY>), reinterpret_cast<const void *>(&tw_pipe_kernel<VAR, 8, LY>), \
                           reinterpret_cast<const void *>(&tw_pipe_kernel<VAR, 4, LY>), reinterpret_cast<const void *>(&tw_pipe_kernel<VAR, 2, LY>)
        const void *kernels[] = {TW_PIPE_K(4, 0), TW_PIPE_K(4, 1), TW_PIPE_K(4, 2), TW_PIPE_K(6, 0), TW_PIPE_K(6, 1), TW_PIPE_K(6, 2)};
#undef TW_PIPE_K
        for (const void *k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIPE_LDS_BYTES) != hipSuccess)
                e->pipeline = 0;
    }
    Params p = base_params(e);
    hipLaunchKernelGGL(tw_pipe_tables_kernel, dim3(1), dim3(64), 0, 0, e->pipe_tab, e->view);
    hipLaunchKernelGGL(tw_reset_kernel, dim3(n_envs), dim3(64), 0, 0, p, (const uint8_t *)nullptr, 0);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipStreamSynchronize(0);
    if (le != hipSuccess) { tw_destroy(e); return hip_fail(le); }
    *out = e;
    return TW_OK;
}

int tw_destroy(tw_engine *e) {
    if (!e) return TW_E_ARG;
    DeviceGuard g(e->device);
    if (e->type) (void)hipFree(e->type);
    if (e->colour) (void)hipFree(e->colour);
    if (e->rec) (void)hipFree(e->rec);
    if (e->type2) (void)hipFree(e->type2);
    if (e->colour2) (void)hipFree(e->colour2);
    if (e->rec2) (void)hipFree(e->rec2);
    if (e->abnormal) (void)hipFree(e->abnormal);
    if (e->pipe_tab) (void)hipFree(e->pipe_tab);
    free(e);
    return TW_OK;
}

int tw_fallback_count(tw_engine *e, int *count) {
    if (!e || !count) return TW_E_ARG;
    DeviceGuard g(e->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(count, e->fb_count, sizeof(int), hipMemcpyDeviceToHost));
    return TW_OK;
}

int tw_set_pipeline(tw_engine *e, int enable) {
    if (!e) return TW_E_ARG;
    e->pipeline = enable ? 1 : 0;
    return TW_OK;
}

int tw_set_envs_per_wave(tw_engine *e, int envs_per_wave) {
    if (!e || !(envs_per_wave == 0 || envs_per_wave == 1 || envs_per_wave == 2 || envs_per_wave == 4)) return TW_E_ARG;
    e->envs_per_wave = envs_per_wave;
    return TW_OK;
}

int tw_reset(tw_engine *e, const uint8_t *mask, uint8_t *obs, int obs_pitch, void *stream) {
    if (!e) return TW_E_ARG;
    DeviceGuard g(e->device);
    Params p = base_params(e);
    p.obs = obs;
    if (obs_pitch > 0) p.obs_pitch = obs_pitch;
    if (p.obs_pitch < e->view * e->view * 3) return TW_E_ARG;
    hipLaunchKernelGGL(tw_reset_kernel, dim3(e->n_envs), dim3(64), 0, (hipStream_t)stream, p, mask, 1);
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

int tw_step(tw_engine *e, const int32_t *actions, const uint32_t *draws, uint8_t *obs, int obs_pitch,
            float *state_matrix, int mat_pitch, float *pos, float *reward, uint8_t *terminated, uint8_t *truncated,
            int flags, void *stream) {
    if (!e || !actions) return TW_E_ARG;
    DeviceGuard g(e->device);
    return launch_rollout(e, 1, actions, draws, obs, obs_pitch, state_matrix, mat_pitch, pos, reward, terminated,
                          truncated, flags, (hipStream_t)stream);
}

int tw_rollout(tw_engine *e, int T, const int32_t *actions, const uint32_t *draws, uint8_t *obs, int obs_pitch,
               float *state_matrix, int mat_pitch, float *pos, float *reward, uint8_t *terminated,
               uint8_t *truncated, int flags, void *stream) {
    if (!e || T <= 0) return TW_E_ARG;
    DeviceGuard g(e->device);
    return launch_rollout(e, T, actions, draws, obs, obs_pitch, state_matrix, mat_pitch, pos, reward, terminated,
                          truncated, flags, (hipStream_t)stream);
}

int tw_fill_actions(tw_engine *e, int T, int32_t *actions, void *stream) {
    if (!e || T <= 0 || !actions) return TW_E_ARG;
    DeviceGuard g(e->device);
    Params p = base_params(e);
    p.T = T;
    const int total = T * e->n_envs;
    hipLaunchKernelGGL(tw_fill_actions_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, p,
                       actions);
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

int tw_state_ptrs(tw_engine *e, uint8_t **type_plane, uint8_t **colour_plane, int32_t **records) {
    if (!e) return TW_E_ARG;
    if (type_plane) *type_plane = e->type;
    if (colour_plane) *colour_plane = e->colour;
    if (records) *records = e->rec;
    return TW_OK;
}

int tw_get_state_host(tw_engine *e, uint8_t *type_plane, uint8_t *colour_plane, int32_t *records) {
    if (!e) return TW_E_ARG;
    DeviceGuard g(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (type_plane) HIP_TRY(hipMemcpy(type_plane, e->type, (size_t)e->n_envs * NC, hipMemcpyDeviceToHost));
    if (colour_plane) HIP_TRY(hipMemcpy(colour_plane, e->colour, (size_t)e->n_envs * NC, hipMemcpyDeviceToHost));
    if (records) HIP_TRY(hipMemcpy(records, e->rec, (size_t)e->n_envs * REC * 4, hipMemcpyDeviceToHost));
    return TW_OK;
}

int tw_set_state_host(tw_engine *e, const uint8_t *type_plane, const uint8_t *colour_plane,
                      const int32_t *records) {
    if (!e) return TW_E_ARG;
    DeviceGuard g(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (type_plane) HIP_TRY(hipMemcpy(e->type, type_plane, (size_t)e->n_envs * NC, hipMemcpyHostToDevice));
    if (colour_plane) HIP_TRY(hipMemcpy(e->colour, colour_plane, (size_t)e->n_envs * NC, hipMemcpyHostToDevice));
    if (records) HIP_TRY(hipMemcpy(e->rec, records, (size_t)e->n_envs * REC * 4, hipMemcpyHostToDevice));
    return TW_OK;
}

int tw_gen_obs(tw_engine *e, int view_size, uint8_t *obs, int obs_pitch, void *stream) {
    if (!e || !obs || !view_ok(view_size)) return TW_E_ARG;
    DeviceGuard g(e->device);
    Params p = base_params(e);
    p.view = view_size; p.obs = obs;
    p.obs_pitch = obs_pitch > 0 ? obs_pitch : view_size * view_size * 3;
    if (p.obs_pitch < view_size * view_size * 3) return TW_E_ARG;
    hipLaunchKernelGGL(tw_gen_obs_kernel, dim3(e->n_envs), dim3(64), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

#ifdef TW_STAMP
int tw_debug_stamps2(unsigned long long *out3072) {
    return hipMemcpyFromSymbol(out3072, HIP_SYMBOL(g_stamp2), sizeof(unsigned long long) * 3072) == hipSuccess ? 0 : -2;
}
int tw_debug_stamps3(unsigned long long *out4096) {
    return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_stamp3), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -2;
}
int tw_debug_stamps(unsigned long long *out512) {
    return hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : -2;
}
#endif

// ---------------------------------------------------------------- engine-owned output slab
namespace {

struct Slab {
    void *base = nullptr;
    size_t bytes = 0, chunk = 0;
    int backing = 0;                                  // 0 hipMalloc, >0 mapped hipMemCreate granules
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> slots;                        // handles[i] is mapped at base + slots[i] * chunk
};
std::mutex g_slab_mu;
std::map<void *, Slab> g_slabs;

int slab_release(Slab &s) {
    if (!s.base) return TW_OK;
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    if (s.backing == 0) { note(hipFree(s.base)); }
    else {
        // one unmap per mapping, then the physical chunks go back to the driver.  The ADDRESS RANGE is deliberately kept
        // reserved (hipMemAddressFree is never called): on ROCm 7.2 a range that was freed and handed out again by a later
        // hipMemAddressReserve served stale translations -- a rollout into the new slab lost ~10 % of its rows
        // (tools/dbg_fallback.py; leak / keep-range / free-range: fine / fine / corrupt).  The cost is virtual
        // address space only (about 1 GiB per released slab of the benchmark size, out of 128 TiB).
        for (size_t i = 0; i < s.handles.size(); ++i) note(hipMemUnmap((char *)s.base + s.slots[i] * s.chunk, s.chunk));
        for (auto h : s.handles) note(hipMemRelease(h));
    }
    s.base = nullptr;
    return first == hipSuccess ? TW_OK : hip_fail(first);
}

// backing: 0 hipMalloc; k > 0: the slab is a contiguous virtual range backed by separately created physical chunks
// (hipMemCreate) of 2^(k-1) x 2 MiB each: 1 -> 2 MiB, 2 -> 4 MiB, 5 -> 32 MiB, ...; 99 -> one chunk for the slab.
// Default 1.  Measured (tools/placement_probe2.py, 4096 x 128 record-layout rollout, 4 slabs each, ms per launch):
// hipMalloc / one chunk / 32-256 MiB chunks 0.190-0.197; 2 MiB chunks mapped in creation order 0.178-0.189; the same
// chunks mapped in reverse order 0.192-0.196 (= the contiguous case: the driver hands out physical memory top-down,
// so reverse order IS physically ascending); shuffled 0.199-0.201.  Why a stream whose 2 MiB pages descend
// physically absorbs these stores ~5 % faster than an ascending one is not understood; it is kept because it is
// reproducible on every box tried and the fallback (hipMalloc) is what every other allocation gets anyway.
// Diagnostic knob TW_SLAB_ORDER: 0 map the chunks in creation order, 1 in reverse order, 2 shuffled.
int slab_alloc(int device, size_t bytes, int backing, Slab &s) {
    s.bytes = bytes; s.backing = backing; s.base = nullptr;
    if (backing == 0) {
        HIP_TRY(hipMalloc(&s.base, bytes));
        return TW_OK;
    }
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
    size_t chunk = backing >= 99 ? bytes : ((size_t)2 << 20) << (backing - 1);
    chunk = (chunk + gran - 1) / gran * gran;
    s.bytes = (bytes + chunk - 1) / chunk * chunk;
    s.chunk = chunk;
    const size_t nchunks = s.bytes / chunk;
    hipError_t err = hipMemAddressReserve(&s.base, s.bytes, gran, nullptr, 0);
    if (err != hipSuccess) { s.base = nullptr; return hip_fail(err); }
    const char *ord_s = getenv("TW_SLAB_ORDER");
    const int order = ord_s ? atoi(ord_s) : 0;
    std::vector<size_t> slot(nchunks);
    for (size_t i = 0; i < nchunks; ++i) slot[i] = order == 1 ? nchunks - 1 - i : i;
    if (order == 2) {
        uint64_t r = 0x9E3779B97F4A7C15ull;
        for (size_t i = nchunks; i > 1; --i) {
            r = r * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(slot[i - 1], slot[(size_t)((r >> 33) % i)]);
        }
    }
    std::vector<char> is_mapped(nchunks, 0);
    for (size_t i = 0; i < nchunks; ++i) {
        hipMemGenericAllocationHandle_t h;
        err = hipMemCreate(&h, chunk, &prop, 0);
        if (err != hipSuccess) break;
        err = hipMemMap((char *)s.base + slot[i] * chunk, chunk, 0, h, 0);
        if (err != hipSuccess) { (void)hipMemRelease(h); break; }
        s.handles.push_back(h);
        s.slots.push_back(slot[i]);
        is_mapped[slot[i]] = 1;
    }
    if (err == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof(acc));
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        err = hipMemSetAccess(s.base, s.bytes, &acc, 1);
    }
    if (err != hipSuccess) {
        for (size_t i = 0; i < nchunks; ++i)
            if (is_mapped[i]) (void)hipMemUnmap((char *)s.base + i * chunk, chunk);
        for (auto h : s.handles) (void)hipMemRelease(h);
        s.handles.clear(); s.slots.clear();
        // the range stays reserved, like in slab_release: a freed range that comes back from a later reserve is the
        // pattern that lost rows (the caller falls back to hipMalloc right after this)
        s.base = nullptr;
        return hip_fail(err);
    }
    return TW_OK;
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

int tw_alloc_outputs(tw_engine *e, int T, int flags, tw_outputs *out) {
    if (!e || !out || T <= 0) return TW_E_ARG;
    DeviceGuard g(e->device);
    memset(out, 0, sizeof(*out));
    const size_t TN = (size_t)T * e->n_envs;
    const bool codes = (flags & TW_F_MATRIX_CODE) != 0;
    const int obs_row = ((e->view * e->view * 3 + 15) >> 4) << 4;
    // ONE stream of records per env-step -- float frames: matrix row (1168 B) | image row, 2048 bytes for V = 17;
    // code frames: image row | code row (304 B).  Two separate streams made the store bandwidth depend on where the
    // driver happened to place them (0.178 ... 0.230 ms per launch).
    const size_t mat_row = codes ? (size_t)MATC_BYTES : (size_t)MAT_WORDS * 4;
    const size_t rec_bytes = mat_row + (size_t)obs_row;
    const size_t A = (size_t)2 << 20;                 // every stream starts on a 2 MiB page
    const size_t sz[6] = {TN * rec_bytes, 0, TN * 8, TN * 4, TN, TN};
    size_t off[6], total = 0;
    for (int i = 0; i < 6; ++i) { off[i] = total; total += round_up(sz[i], A); }
    // backing: mapped 2 MiB granules by default (TW_SLAB_BACKING=0..3 overrides); hipMalloc when the runtime refuses
    const char *bk = getenv("TW_SLAB_BACKING");
    int backing = bk ? atoi(bk) : e->slab_backing;
    if (flags & TW_F_SLAB_HIPMALLOC) backing = 0;
    Slab s;
    int rc = slab_alloc(e->device, total, backing, s);
    if (rc != TW_OK && backing != 0) { backing = 0; rc = slab_alloc(e->device, total, 0, s); }
    if (rc != TW_OK) return rc;
    char *b = (char *)s.base;
    out->matrix = codes ? b + off[0] + obs_row : b + off[0];
    out->obs = (uint8_t *)(codes ? b + off[0] : b + off[0] + REC_OBS_OFF);
    out->pos = (float *)(b + off[2]);
    out->reward = (float *)(b + off[3]); out->terminated = (uint8_t *)(b + off[4]); out->truncated = (uint8_t *)(b + off[5]);
    out->obs_pitch = (int)rec_bytes; out->mat_pitch = codes ? (int)rec_bytes : (int)(rec_bytes / 4);
    out->T = T; out->n_envs = e->n_envs; out->flags = flags & TW_F_MATRIX_CODE; out->device = e->device;
    out->backing = s.backing; out->slab_bytes = s.bytes; out->slab = s.base;
    std::lock_guard<std::mutex> lk(g_slab_mu);
    g_slabs[s.base] = s;
    return TW_OK;
}

int tw_free_outputs(tw_outputs *out) {
    if (!out || !out->slab) return TW_E_ARG;
    DeviceGuard g(out->device);
    Slab s;
    {
        std::lock_guard<std::mutex> lk(g_slab_mu);
        auto it = g_slabs.find(out->slab);
        if (it == g_slabs.end()) return TW_E_ARG;
        s = it->second;
        g_slabs.erase(it);
    }
    HIP_TRY(hipDeviceSynchronize());
    const int rc = slab_release(s);
    memset(out, 0, sizeof(*out));
    return rc;
}

int tw_abi_sizeof_outputs(void) { return (int)sizeof(tw_outputs); }

int tw_n_envs(const tw_engine *e) { return e ? e->n_envs : TW_E_ARG; }
int tw_view_size(const tw_engine *e) { return e ? e->view : TW_E_ARG; }
int tw_last_hip_error(void) { return g_last_hip_error; }
const char *tw_last_error_message(void) { return g_last_error_msg; }
const char *tw_version(void) { return "twoarmy-hip 0.3 (gfx950)"; }
#ifndef TW_BUILD_ID
#define TW_BUILD_ID "unknown"
#endif
const char *tw_build_id(void) { return TW_BUILD_ID; }

int tw_time_rollout(tw_engine *e, int T, const int32_t *actions, uint8_t *obs, int obs_pitch, float *state_matrix,
                    int mat_pitch, float *pos, float *reward, uint8_t *terminated, uint8_t *truncated, int flags,
                    int iters, void *stream, float *ms_per_launch) {
    if (!e || T <= 0 || iters <= 0 || !ms_per_launch) return TW_E_ARG;
    DeviceGuard g(e->device);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a = nullptr, b = nullptr;
    HIP_TRY(hipEventCreate(&a));
    hipError_t he = hipEventCreate(&b);
    int rc = TW_OK;
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventRecord(a, st);
    for (int i = 0; he == hipSuccess && rc == TW_OK && i < iters; ++i)
        rc = launch_rollout(e, T, actions, nullptr, obs, obs_pitch, state_matrix, mat_pitch, pos, reward,
                            terminated, truncated, flags, st);
    if (he == hipSuccess && rc == TW_OK) he = hipEventRecord(b, st);
    if (he == hipSuccess && rc == TW_OK) he = hipEventSynchronize(b);
    if (he == hipSuccess && rc == TW_OK) he = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);                         // both events are released on every path
    if (b) (void)hipEventDestroy(b);
    if (rc != TW_OK) return rc;
    if (he != hipSuccess) return hip_fail(he);
    *ms_per_launch = ms / (float)iters;
    return TW_OK;
}

}  // extern "C"
