// twoarmy_engine.hip -- MI355X (gfx950) MiniGrid-Twoarmy step / observation engine.
//
// One wavefront (= one 64-thread workgroup) owns one environment.  The env's object planes
// (type, colour: SoA uint8[N][289] in HBM) are staged into LDS as packed 32-bit cells once per
// launch and stay there for all T steps of a rollout; the per-env scalar record is held in
// wave-uniform registers (env index = blockIdx.x, so hipcc keeps the step logic on the scalar
// unit).  Per step a wave
//   1. applies the uniform transition logic (ball/patrol moves, agent move, termination),
//   2. gathers the V x V egocentric window from LDS (closed-form rotation, OOB -> wall),
//      stages the uint8[V][V][3] image in LDS and streams it out as aligned dwords,
//   3. applies the post-observation logic (wall drop, patrol spawn, shaped reward, risk counter,
//      episode-end re-arm),
//   4. emits the fp32 289-cell state matrix, agent (y,x), reward, terminated, truncated,
//   5. optionally re-generates the grid in place (auto-reset).
//
// Behavioural contract (bit-exact vs the reference, checked against oracle/ + tests/golden):
//   gym_minigrid/envs/twoarmy_v6.py:83-325, twoarmy_v4.py:82-322  (Twoarmy step)
//   gym_minigrid/minigrid.py:1333-1441 (MiniGridEnv.step), :1262-1293, :1443-1496, :641-660,
//   :627-639, :749-772 (view extents, slice, rotate_left, encode)
//   soa/env_buffer.py:300-334 (matrix_env, data_env)
// This is a from-scratch closed-form implementation; it shares no code with oracle/.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "twoarmy.h"

namespace {

constexpr int GS = TW_GRID;
constexpr int NC = TW_CELLS;
constexpr int REC = TW_REC_WORDS;

// packed LDS cell: type | colour << 8 | state << 16   (OBJECT_TO_IDX / COLOR_TO_IDX, minigrid.py:40-67)
constexpr uint32_t C_EMPTY = 1u;
constexpr uint32_t C_WALL = 2u | (5u << 8);
constexpr uint32_t C_BALL = 6u | (4u << 8);
constexpr uint32_t C_GOAL = 8u | (1u << 8);

enum { R_STEP = 0, R_RISK = 1, R_HIT = 2, R_ROOM2 = 3, R_GOAL = 4 };

struct Params {
    uint8_t *type;
    uint8_t *colour;
    int32_t *rec;
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
    int flags;
};

// ---------------------------------------------------------------- Philox4x32-10 (uniform -> SALU)
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

__device__ __forceinline__ uint32_t draw_word(uint32_t k0, uint32_t k1, uint32_t env_id, uint32_t t,
                                              uint32_t slot) {
    uint32_t c0 = env_id, c1 = t, c2 = slot >> 2, c3 = 0x54574F41u;
    philox4x32_10(k0, k1, c0, c1, c2, c3);
    const uint32_t w = slot & 3u;
    return w == 0 ? c0 : (w == 1 ? c1 : (w == 2 ? c2 : c3));
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

// wave-uniform env scalars (names follow the reference attributes)
struct EnvS {
    int ax, ay, dir, step_count, step_move, pone, patrol, up1, right2, upd_long, upd_horiz, risk,
        first_room2;
    int obx[3], oby[3], o1x[3], o1y[3], o1v, o2x[4], o2y[4], o2v, gx, gy;
    uint32_t t;
    int err, max_steps, episodes, last_reward, last_term, last_trunc;
};

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One workgroup == one wavefront: LDS operations of a wave execute in issue order, so cross-lane
// LDS hand-offs only need the COMPILER to keep program order.  __syncthreads() would also emit
// s_waitcnt vmcnt(0) and drain every outstanding global store (several microseconds per step under load).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void load_env(EnvS &s, const int32_t *r) {
    s.ax = rfl(r[TW_AX]); s.ay = rfl(r[TW_AY]); s.dir = rfl(r[TW_DIR]);
    s.step_count = rfl(r[TW_STEP_COUNT]); s.step_move = rfl(r[TW_STEP_MOVE]);
    s.pone = rfl(r[TW_PONE]); s.patrol = rfl(r[TW_PATROL]); s.up1 = rfl(r[TW_UP1]);
    s.right2 = rfl(r[TW_RIGHT2]); s.upd_long = rfl(r[TW_UPD_LONG]); s.upd_horiz = rfl(r[TW_UPD_HORIZ]);
    s.risk = rfl(r[TW_RISK]); s.first_room2 = rfl(r[TW_FIRST_ROOM2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.obx[k] = rfl(r[TW_OBX + k]); s.oby[k] = rfl(r[TW_OBY + k]);
        s.o1x[k] = rfl(r[TW_O1X + k]); s.o1y[k] = rfl(r[TW_O1Y + k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.o2x[k] = rfl(r[TW_O2X + k]); s.o2y[k] = rfl(r[TW_O2Y + k]); }
    s.o1v = rfl(r[TW_O1_VALID]); s.o2v = rfl(r[TW_O2_VALID]);
    s.gx = rfl(r[TW_GOAL_X]); s.gy = rfl(r[TW_GOAL_Y]);
    s.t = (uint32_t)rfl(r[TW_T]); s.err = rfl(r[TW_ERROR]); s.max_steps = rfl(r[TW_MAX_STEPS]);
    s.episodes = rfl(r[TW_EPISODES]); s.last_reward = rfl(r[TW_LAST_REWARD]);
    s.last_term = rfl(r[TW_LAST_TERM]); s.last_trunc = rfl(r[TW_LAST_TRUNC]);
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
}

// MiniGridEnv.reset + _gen_grid on the scalar side (minigrid.py:947-980, twoarmy_v6.py:56-77).
__device__ __forceinline__ void reset_scalars(EnvS &s) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.obx[k] = 7 + k; s.oby[k] = 8; }
    s.o1v = 0; s.o2v = 0;
    s.ax = 3; s.ay = 15; s.dir = 3; s.gx = 14; s.gy = 2;
    s.step_count = 0; s.err = 0;
}

__device__ __forceinline__ bool inb(int x, int y) { return (unsigned)x < (unsigned)GS && (unsigned)y < (unsigned)GS; }

// Grid.set by one lane (minigrid.py:599-602); false where the reference's bounds assert fires.
__device__ __forceinline__ bool set_cell(uint32_t *cells, int lane, int x, int y, uint32_t v) {
    if (!inb(x, y)) return false;
    if (lane == 0) cells[y * GS + x] = v;
    return true;
}

// Patrol group move: clear every cell, then put each ball at +d inside try/except
// (twoarmy_v4.py:119-176).  n <= 4.
template <int NB>
__device__ __forceinline__ bool move_group(uint32_t *cells, int lane, int (&xs)[NB], int (&ys)[NB], int valid,
                                           int dx, int dy, int &err) {
    if (!valid) { err = TW_ENV_TYPE; return false; }
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (!set_cell(cells, lane, xs[k], ys[k], C_EMPTY)) { err = TW_ENV_ASSERT; return false; }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int nx = xs[k] + dx, ny = ys[k] + dy;
        if (set_cell(cells, lane, nx, ny, C_BALL)) { xs[k] = nx; ys[k] = ny; }
    }
    return true;
}

// gen_obs_grid + encode (minigrid.py:1443-1496) in closed form.
//   view cell (i, j) (i = x index of the rotated view, j = y index; output byte (i*V + j)*3 + ch)
//   maps to slice coords   dir 0: (V-1-j, i)   1: (V-1-i, V-1-j)   2: (j, V-1-i)   3: (i, j)
//   and slice origin       dir 0: (ax, ay-h)   1: (ax-h, ay)       2: (ax-V+1, ay-h)   3: (ax-h, ay-V+1)
// Out-of-grid cells read as Wall (2,5,0) (minigrid.py:655-656); the agent's own cell (V/2, V-1) is
// forced empty (minigrid.py:1472-1476, carrying is always None in Twoarmy).
// The image is staged in LDS at the same 4-byte phase as its global destination and copied out as
// aligned dwords (+ <= 3 head / tail bytes).
struct ViewIdx { int i[5], j[5]; };   // per-lane view coords of cells lane + 64k (hoisted out of the step loop)

__device__ __forceinline__ ViewIdx make_view_idx(int lane, int V) {
    ViewIdx w;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        w.i[k] = c / V;
        w.j[k] = c - w.i[k] * V;
    }
    return w;
}

__device__ __forceinline__ void emit_obs(const uint32_t *cells, uint32_t *stage, int lane, int V, int ax, int ay,
                                         int dir, uint8_t *dst, const ViewIdx &w) {
    const int VV = V * V, h = V >> 1, nb = VV * 3;
    int topx, topy;
    if (dir == 0) { topx = ax; topy = ay - h; }
    else if (dir == 1) { topx = ax - h; topy = ay; }
    else if (dir == 2) { topx = ax - V + 1; topy = ay - h; }
    else { topx = ax - h; topy = ay - V + 1; }
    const int c_agent = h * V + V - 1;
    const uint32_t off = (uint32_t)(uintptr_t)dst & 3u;
    uint8_t *sb = reinterpret_cast<uint8_t *>(stage) + off;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        if (c >= VV) break;
        const int i = w.i[k], j = w.j[k];
        int sx, sy;
        if (dir == 0) { sx = V - 1 - j; sy = i; }
        else if (dir == 1) { sx = V - 1 - i; sy = V - 1 - j; }
        else if (dir == 2) { sx = j; sy = V - 1 - i; }
        else { sx = i; sy = j; }
        const int x = topx + sx, y = topy + sy;
        uint32_t v = inb(x, y) ? cells[y * GS + x] : C_WALL;
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

// Env_transact.matrix_env (soa/env_buffer.py:300-318)
__device__ __forceinline__ void emit_matrix(const uint32_t *cells, int lane, int ax, int ay, float *dst) {
    const int ca = ay * GS + ax;
    for (int c = lane; c < NC; c += 64) {
        const uint32_t ty = cells[c] & 0xffu;
        float m = ty == 2u ? -0.9f : (ty == 6u ? -0.5f : 0.9f);
        if (c == ca) m = 0.3f;
        dst[c] = m;
    }
}

__device__ __forceinline__ float reward_value(int code) {
    return code == R_STEP ? -0.01f : code == R_RISK ? -0.1f : code == R_HIT ? -0.9f : code == R_ROOM2 ? 0.2f : 0.9f;
}

// ---------------------------------------------------------------- the rollout kernel
__global__ __launch_bounds__(64) void tw_rollout_kernel(Params p) {
    __shared__ uint32_t cells[NC + 3];
    __shared__ uint32_t stage[(NC * 3 + 8) / 4 + 1];
    __shared__ int32_t recs[REC];

    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = p.n_envs;
    const uint32_t env_id = p.env_id0 + (uint32_t)n;

    if (lane < REC) recs[lane] = p.rec[(size_t)n * REC + lane];
    for (int c = lane; c < NC; c += 64)
        cells[c] = (uint32_t)p.type[(size_t)n * NC + c] | ((uint32_t)p.colour[(size_t)n * NC + c] << 8);
    wave_sync();
    EnvS s;
    load_env(s, recs);

    const int V = p.view;
    const ViewIdx widx = make_view_idx(lane, V);
    const size_t obs_bytes = (size_t)V * V * 3;
    const bool v4 = p.variant == 4;
    const bool autoreset = (p.flags & TW_F_AUTORESET) != 0;
    const bool policy_idx = (p.flags & TW_F_POLICY_IDX) != 0 || p.actions == nullptr;

    int act_vec = 0;                                  // lane l: action of step (tt & ~63) + l
    for (int tt = 0; tt < p.T; ++tt) {
        const size_t idx = (size_t)tt * N + n;
        if (p.actions && (tt & 63) == 0) {
            const int ts = tt + lane;
            act_vec = ts < p.T ? p.actions[(size_t)ts * N + n] : 0;
        }
        const uint32_t t = s.t;
        const uint32_t *dr = p.draws ? p.draws + idx * TW_DRAW_WORDS : nullptr;
        auto take = [&](uint32_t slot) -> uint32_t {
            return dr ? (uint32_t)rfl((int)dr[slot]) : draw_word(p.seed_lo, p.seed_hi, env_id, t, slot);
        };

        int action;
        if (p.actions) action = __builtin_amdgcn_readlane(act_vec, tt & 63);
        else action = (int)(draw_word(p.seed_lo, p.seed_hi, env_id, t, TW_S_ACTION) % 5u);
        if (policy_idx && action == 4) action = 6;                // Env_transact.env_action

        // ================= part 1: pre-observation transition
        s.t += 1;
        int err = TW_ENV_OK;
        int terminated = 0, truncated = 0, reward = R_STEP;
        bool have_obs = false;
        do {
            if (action >= 7) action = 0;                          // twoarmy_v6.py:85-86
            s.step_move += 1;                                     // :88
            const int sm = s.step_move;
            const int m6 = sm % 6;
            // row-8 balls (:96-112): clear all, then put each inside try/except
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (ok && !set_cell(cells, lane, s.obx[k], s.oby[k], C_EMPTY)) ok = false;
            if (!ok) { err = TW_ENV_ASSERT; break; }
            const int dxb = (m6 == 1 || m6 == 0) ? 1 : ((m6 == 2 || m6 == 3) ? -1 : 0);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int nx = s.obx[k] + dxb;
                if (set_cell(cells, lane, nx, 8, C_BALL)) { s.obx[k] = nx; s.oby[k] = 8; }
            }
            if (v4) {
                if (s.upd_long) {                                 // twoarmy_v4.py:115-144
                    s.upd_horiz = 0;
                    bool go = (sm % 4 == 2) || (m6 == 3) || (m6 == 0);
                    if (!go) go = (take(TW_S_GATE) % 10u) == 6u;
                    if (go && s.patrol) {
                        if (s.up1) {
                            if (!move_group<3>(cells, lane, s.o1x, s.o1y, s.o1v, 0, -1, err)) break;
                            if (s.o1y[0] == 3) s.up1 = 0;
                        } else {
                            if (!move_group<3>(cells, lane, s.o1x, s.o1y, s.o1v, 0, 1, err)) break;
                            if (s.o1y[2] == 7) s.up1 = 1;
                        }
                    }
                }
                if (s.upd_horiz) {                                // twoarmy_v4.py:147-176
                    s.upd_long = 0;
                    bool go = (m6 != 1);
                    if (!go) go = (take(TW_S_GATE) % 10u) == 6u;
                    if (go && s.patrol) {
                        if (s.right2) {
                            if (!move_group<4>(cells, lane, s.o2x, s.o2y, s.o2v, 1, 0, err)) break;
                            if (s.o2x[3] == 11) s.right2 = 0;
                        } else {
                            if (!move_group<4>(cells, lane, s.o2x, s.o2y, s.o2v, -1, 0, err)) break;
                            if (s.o2x[0] == 5) s.right2 = 1;
                        }
                    }
                }
            }
            // ---- MiniGridEnv.step (minigrid.py:1333-1441)
            s.step_count += 1;
            {   // front_pos is read for every action (:1341-1344): bounds assert
                const int fx = s.ax + (s.dir == 0 ? 1 : (s.dir == 2 ? -1 : 0));
                const int fy = s.ay + (s.dir == 1 ? 1 : (s.dir == 3 ? -1 : 0));
                if (!inb(fx, fy)) { err = TW_ENV_ASSERT; break; }
            }
            int tx = s.ax, ty = s.ay;
            if (action == 0) tx -= 1;
            else if (action == 1) tx += 1;
            else if (action == 2) ty -= 1;
            else if (action == 3) ty += 1;
            else if (action != 6) { err = TW_ENV_ATTRIBUTE; break; }   // self.actions.forward, :1397
            if (!inb(tx, ty)) { err = TW_ENV_ASSERT; break; }
            wave_sync();                                      // lane-0 cell writes -> all lanes
            {
                const uint32_t cv = (uint32_t)rfl((int)cells[ty * GS + tx]);
                const uint32_t ct = cv & 0xffu, cs = (cv >> 16) & 0xffu;
                const bool overlap = ct == 8u || ct == 11u || ct == 3u || ct == 9u || (ct == 4u && cs == 0u);
                if (ct == 1u || overlap) { s.ax = tx; s.ay = ty; }
                if (ct == 8u) terminated = 1;
            }
            if (s.step_count >= s.max_steps) truncated = 1;       // :1436-1437
            have_obs = true;
        } while (false);

        if (!have_obs) {           // the reference raised: state keeps the mutations made so far
            s.err = err;
            s.last_reward = -1; s.last_term = 0; s.last_trunc = 0;
            wave_sync();
            continue;
        }

        // ================= observation (before wall drop / spawn of the same step)
        if (p.obs) emit_obs(cells, stage, lane, V, s.ax, s.ay, s.dir, p.obs + idx * obs_bytes, widx);

        // ================= part 2: post-observation logic
        do {
            if (!s.pone && (s.ax > 3 || s.ay < 14)) {             // twoarmy_v6.py:182-198 / v4:181-195
                int i1 = 11, i2 = 8;
                if (v4) {
                    i1 = 9 + (int)(take(TW_S_WALL1) % 4u);
                    i2 = 6 + (int)(take(TW_S_WALL2) % 4u);
                }
                if (lane < 8) {
                    const int q = lane & 3, blk = lane >> 2;       // 2x2 blocks
                    const int x = blk == 0 ? 4 + (q & 1) : i2 + (q >> 1);
                    const int y = blk == 0 ? i1 + (q >> 1) : 11 + (q & 1);
                    cells[y * GS + x] = C_WALL;
                }
                s.pone = 1;
            }
            if (v4 && !s.patrol && s.ay <= 8) {                   // twoarmy_v4.py:212-225
                const int i = 6 + (int)(take(TW_S_SPAWN) % 4u);
                s.o2x[0] = i; s.o2x[1] = i + 1; s.o2x[2] = i; s.o2x[3] = i + 1;
                s.o2y[0] = 4; s.o2y[1] = 4; s.o2y[2] = 5; s.o2y[3] = 5;
#pragma unroll
                for (int k = 0; k < 3; ++k) { s.o1x[k] = 12; s.o1y[k] = 4 + k; }
                if (lane < 7) {
                    const int x = lane < 4 ? i + (lane & 1) : 12;
                    const int y = lane < 4 ? 4 + (lane >> 1) : lane;   // lanes 4,5,6 -> y 4,5,6
                    cells[y * GS + x] = C_BALL;
                }
                s.o1v = 1; s.o2v = 1; s.patrol = 1;
            }
            // row-ball collision / proximity (twoarmy_v6.py:231-243)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (s.ax == s.obx[k] && s.ay == s.oby[k]) { reward = R_HIT; truncated = 1; }
            if (s.ay == s.oby[0] + 1 && (s.ax == s.obx[0] || s.ax == s.obx[1] || s.ax == s.obx[2])) reward = R_RISK;
            if (s.patrol) {                                       // :245-283
                if (!s.o1v || !s.o2v) { err = TW_ENV_TYPE; break; }
                if (s.ay == s.o2y[2] + 1 && (s.ax == s.o2x[2] || s.ax == s.o2x[3])) reward = R_RISK;
                if (s.ax == s.o2x[0] - 1 && (s.ay == s.o2y[0] || s.ay == s.o2y[2])) reward = R_RISK;
                if (s.ax == s.o2x[1] + 1 && (s.ay == s.o2y[1] || s.ay == s.o2y[3])) reward = R_RISK;
                if (s.ax == s.o1x[0] - 1 && (s.ay == s.o1y[0] || s.ay == s.o1y[1] || s.ay == s.o1y[2])) reward = R_RISK;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (s.ax == s.o1x[k] && s.ay == s.o1y[k]) { reward = R_HIT; truncated = 1; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (s.ax == s.o2x[k] && s.ay == s.o2y[k]) { reward = R_HIT; truncated = 1; }
            }
            if (s.first_room2 && s.ay == 7) { reward = R_ROOM2; s.first_room2 = 0; }   // :285-288
            if (reward == R_RISK) {                               // :290-294
                s.risk += 1;
                if (s.risk > 5) truncated = 1;
            }
            if (terminated || truncated) {                        // :296-318
                if (terminated) reward = R_GOAL;
                s.step_move = 0; s.pone = 0; s.patrol = 0; s.first_room2 = 1; s.risk = 0;
                if ((take(TW_S_COIN_A) & 1u) == 1u) { s.up1 = 0; s.right2 = 1; } else { s.up1 = 1; s.right2 = 0; }
                if ((take(TW_S_COIN_B) & 1u) == 1u) { s.upd_horiz = 0; s.upd_long = 1; } else { s.upd_horiz = 1; s.upd_long = 0; }
                s.episodes += 1;
            }
        } while (false);
        s.err = err;
        if (err != TW_ENV_OK) {
            s.last_reward = -1; s.last_term = 0; s.last_trunc = 0;
            wave_sync();
            continue;
        }
        s.last_reward = reward; s.last_term = terminated; s.last_trunc = truncated;
        wave_sync();                                          // wall / spawn cell writes -> all lanes

        // ================= outputs after the full step
        if (p.matrix) emit_matrix(cells, lane, s.ax, s.ay, p.matrix + idx * NC);
        if (lane == 0) {
            if (p.reward) p.reward[idx] = reward_value(reward);
            if (p.term) p.term[idx] = (uint8_t)terminated;
            if (p.trunc) p.trunc[idx] = (uint8_t)truncated;
        }
        if (p.pos && lane < 2) p.pos[idx * 2 + lane] = (float)(lane == 0 ? s.ay : s.ax);

        // ================= auto-reset (soa/train_ppo.py:104: reset() opens every episode)
        if (autoreset && (terminated || truncated)) {
            wave_sync();
            for (int c = lane; c < NC; c += 64) {
                const int y = c / GS, x = c - y * GS;
                cells[c] = gen_cell(x, y);
            }
            reset_scalars(s);
        }
        wave_sync();
    }

    // write state back
    if (lane == 0) store_env(s, recs);
    wave_sync();
    if (lane < REC) p.rec[(size_t)n * REC + lane] = recs[lane];
    for (int c = lane; c < NC; c += 64) {
        const uint32_t v = cells[c];
        p.type[(size_t)n * NC + c] = (uint8_t)v;
        p.colour[(size_t)n * NC + c] = (uint8_t)(v >> 8);
    }
}

// ---------------------------------------------------------------- init / reset / obs-only kernels
// mode 0: Twoarmy __init__ (flags armed, twoarmy_v6.py:15-25) + reset;  mode 1: MiniGridEnv.reset only.
__global__ __launch_bounds__(64) void tw_reset_kernel(Params p, const uint8_t *mask, int mode) {
    __shared__ uint32_t cells[NC + 3];
    __shared__ uint32_t stage[(NC * 3 + 8) / 4 + 1];
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
    __syncthreads();
    if (p.obs) emit_obs(cells, stage, lane, p.view, 3, 15, 3, p.obs + (size_t)n * p.view * p.view * 3,
                        make_view_idx(lane, p.view));
}

__global__ __launch_bounds__(64) void tw_gen_obs_kernel(Params p) {
    __shared__ uint32_t cells[NC + 3];
    __shared__ uint32_t stage[(NC * 3 + 8) / 4 + 1];
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int c = lane; c < NC; c += 64)
        cells[c] = (uint32_t)p.type[(size_t)n * NC + c] | ((uint32_t)p.colour[(size_t)n * NC + c] << 8);
    __syncthreads();
    const int32_t *r = p.rec + (size_t)n * REC;
    emit_obs(cells, stage, lane, p.view, rfl(r[TW_AX]), rfl(r[TW_AY]), rfl(r[TW_DIR]),
             p.obs + (size_t)n * p.view * p.view * 3, make_view_idx(lane, p.view));
}

__global__ void tw_fill_actions_kernel(Params p, int32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = p.T * p.n_envs;
    if (i >= total) return;
    const int tt = i / p.n_envs, n = i - tt * p.n_envs;
    const uint32_t t = (uint32_t)p.rec[(size_t)n * REC + TW_T] + (uint32_t)tt;
    out[i] = (int32_t)(draw_word(p.seed_lo, p.seed_hi, p.env_id0 + (uint32_t)n, t, TW_S_ACTION) % 5u);
}

int g_last_hip_error = 0;
char g_last_error_msg[256] = "";

}  // namespace

// ======================================================================= host side / C ABI
struct tw_engine {
    int variant, n_envs, view, device;
    uint64_t seed;
    uint32_t env_id0;
    uint8_t *type, *colour;
    int32_t *rec;
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
    p.n_envs = e->n_envs; p.view = e->view; p.variant = e->variant;
    p.seed_lo = (uint32_t)e->seed; p.seed_hi = (uint32_t)(e->seed >> 32);
    p.env_id0 = e->env_id0;
    p.T = 1;
    return p;
}

bool view_ok(int v) { return v >= 3 && v <= GS && (v & 1); }

int launch_rollout(tw_engine *e, int T, const int32_t *actions, const uint32_t *draws, uint8_t *obs,
                   float *matrix, float *pos, float *reward, uint8_t *term, uint8_t *trunc, int flags,
                   hipStream_t st) {
    Params p = base_params(e);
    p.T = T; p.actions = actions; p.draws = draws; p.obs = obs; p.matrix = matrix; p.pos = pos;
    p.reward = reward; p.term = term; p.trunc = trunc; p.flags = flags;
    hipLaunchKernelGGL(tw_rollout_kernel, dim3(e->n_envs), dim3(64), 0, st, p);
    HIP_TRY(hipGetLastError());
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
    hipError_t r1 = hipMalloc((void **)&e->type, (size_t)n_envs * NC);
    hipError_t r2 = hipMalloc((void **)&e->colour, (size_t)n_envs * NC);
    hipError_t r3 = hipMalloc((void **)&e->rec, (size_t)n_envs * REC * sizeof(int32_t));
    if (r1 != hipSuccess || r2 != hipSuccess || r3 != hipSuccess) {
        if (e->type) (void)hipFree(e->type);
        if (e->colour) (void)hipFree(e->colour);
        if (e->rec) (void)hipFree(e->rec);
        free(e);
        return hip_fail(r1 != hipSuccess ? r1 : (r2 != hipSuccess ? r2 : r3));
    }
    Params p = base_params(e);
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
    (void)hipFree(e->type); (void)hipFree(e->colour); (void)hipFree(e->rec);
    free(e);
    return TW_OK;
}

int tw_reset(tw_engine *e, const uint8_t *mask, uint8_t *obs, void *stream) {
    if (!e) return TW_E_ARG;
    DeviceGuard g(e->device);
    Params p = base_params(e);
    p.obs = obs;
    hipLaunchKernelGGL(tw_reset_kernel, dim3(e->n_envs), dim3(64), 0, (hipStream_t)stream, p, mask, 1);
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

int tw_step(tw_engine *e, const int32_t *actions, const uint32_t *draws, uint8_t *obs, float *state_matrix,
            float *pos, float *reward, uint8_t *terminated, uint8_t *truncated, int flags, void *stream) {
    if (!e || !actions) return TW_E_ARG;
    DeviceGuard g(e->device);
    return launch_rollout(e, 1, actions, draws, obs, state_matrix, pos, reward, terminated, truncated, flags,
                          (hipStream_t)stream);
}

int tw_rollout(tw_engine *e, int T, const int32_t *actions, const uint32_t *draws, uint8_t *obs,
               float *state_matrix, float *pos, float *reward, uint8_t *terminated, uint8_t *truncated,
               int flags, void *stream) {
    if (!e || T <= 0) return TW_E_ARG;
    DeviceGuard g(e->device);
    return launch_rollout(e, T, actions, draws, obs, state_matrix, pos, reward, terminated, truncated, flags,
                          (hipStream_t)stream);
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

int tw_gen_obs(tw_engine *e, int view_size, uint8_t *obs, void *stream) {
    if (!e || !obs || !view_ok(view_size)) return TW_E_ARG;
    DeviceGuard g(e->device);
    Params p = base_params(e);
    p.view = view_size; p.obs = obs;
    hipLaunchKernelGGL(tw_gen_obs_kernel, dim3(e->n_envs), dim3(64), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return TW_OK;
}

int tw_n_envs(const tw_engine *e) { return e ? e->n_envs : TW_E_ARG; }
int tw_view_size(const tw_engine *e) { return e ? e->view : TW_E_ARG; }
int tw_last_hip_error(void) { return g_last_hip_error; }
const char *tw_last_error_message(void) { return g_last_error_msg; }
const char *tw_version(void) { return "twoarmy-hip 0.1 (gfx950)"; }

int tw_time_rollout(tw_engine *e, int T, const int32_t *actions, uint8_t *obs, float *state_matrix, float *pos,
                    float *reward, uint8_t *terminated, uint8_t *truncated, int flags, int iters, void *stream,
                    float *ms_per_launch) {
    if (!e || T <= 0 || iters <= 0 || !ms_per_launch) return TW_E_ARG;
    DeviceGuard g(e->device);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) {
        int rc = launch_rollout(e, T, actions, nullptr, obs, state_matrix, pos, reward, terminated, truncated,
                                flags, st);
        if (rc != TW_OK) return rc;
    }
    HIP_TRY(hipEventRecord(b, st));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = ms / (float)iters;
    return TW_OK;
}

}  // extern "C"
