// minigrid_view.hip -- general MiniGrid agent view for MI355X (gfx950), C ABI in include/minigrid_view.h.
//
// gen_obs_grid + process_vis + encode of gym_minigrid/minigrid.py:1443-1496 / :795-832 / :749-772 for N envs
// with arbitrary W x H worlds, agent direction, view size, carried object and occlusion.
//
// One wavefront serves E = 64 / V envs.  Each V x V window is gathered straight into its rotated position (the
// dir+1 rotate_left calls of the reference are a closed-form index map) into LDS.  process_vis is a row-sequential
// flood; within a row both of its sweeps are carry chains, so a row of ALL E envs is resolved with 64-bit mask
// arithmetic on the scalar unit: the "see behind" bits come from one ballot (env e owns bits e*V .. e*V+V-1), the
// left-to-right sweep is r[i] = m[i] | (r[i-1] & p[i-1]) (one 64-bit add: the carry IS the flood; cut at env boundaries), the
// right-to-left sweep its mirror image, and the seeds of the next row are shifts of (reached & see-behind).
// The encoded images are staged in LDS in output order and leave as aligned dwords.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "minigrid_view.h"
#include "twoarmy.h"

namespace {

constexpr uint32_t T_EMPTY = 1, T_WALL = 2, T_DOOR = 4;
constexpr uint32_t WALL_CELL = 2u | (5u << 8);                 // Wall().encode() = (2, grey 5, 0)
constexpr uint32_t EMPTY_CELL = 1u;                            // encode of None = (1, 0, 0)

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bool see_behind(uint32_t cell) {     // Wall: never; Door: only when open (state 0)
    const uint32_t t = cell & 0xffu;
    return t == T_WALL ? false : (t == T_DOOR ? ((cell >> 16) & 0xffu) == 0u : true);
}

// r[i] = m[i] | (r[i-1] & q[i]): bits of m flood to the right through set bits of q (q[i] = "cell i-1 lets light
// through and cell i belongs to the same env").  One 64-bit ADD does the whole flood: with R = q | m, the carry that
// m injects at a seed ripples up through the consecutive ones of R above it and stops at the first zero, so the bits
// that differ between R + m and R are exactly the flooded run (plus the terminating zero, masked off by R).  A second
// seed inside a run keeps its bit in the sum (1 + 1 + carry) and passes the carry on: `| m` puts it back.  A carry that
// crosses into the next env's segment can only pass a bit that is itself a seed there, and whatever lies above a seed
// is flooded by that seed anyway.  5 scalar instructions instead of a 5-step Kogge-Stone (25).
__device__ __forceinline__ uint64_t flood_right(uint64_t m, uint64_t q) {
    const uint64_t R = q | m;
    return (((R + m) ^ R) & R) | m;
}

// r[i] = m[i] | (r[i+1] & q[i]): the mirror image, through a bit reversal (s_brev_b64)
__device__ __forceinline__ uint64_t flood_left(uint64_t m, uint64_t q) {
    return __builtin_bitreverse64(flood_right(__builtin_bitreverse64(m), __builtin_bitreverse64(q)));
}

// One wavefront serves E = 64 / V envs: a view row of all of them sits in one 64-bit mask (env e owns bits
// e*V .. e*V+V-1), so one ballot + one pair of floods resolves that row of process_vis for E envs at once.
template <int VT>                                                          // VT > 0: view size known at compile time
__global__ __launch_bounds__(64) void mg_gen_obs_kernel(const uint8_t *__restrict__ type, const uint8_t *__restrict__ colour,
                                                        const uint8_t *__restrict__ state, int N, int W, int H,
                                                        const int32_t *__restrict__ agent_x,
                                                        const int32_t *__restrict__ agent_y,
                                                        const int32_t *__restrict__ agent_dir,
                                                        const uint8_t *__restrict__ carrying, int v_rt, int see_through,
                                                        uint8_t *__restrict__ image, int image_pitch,
                                                        uint8_t *__restrict__ vis_mask) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;
    const int V = VT ? VT : v_rt, E = 64 / V;
    const int n0 = blockIdx.x * E;
    const int VV = V * V, EV = E * V, nb = VV * 3, nbp = (nb + 6) & ~3;     // nbp: staged bytes per env incl. alignment phase
    const int ne = N - n0 < E ? N - n0 : E;                                 // envs of this wavefront
    uint32_t *cells = lds;                                                  // [e][j][i]: type | colour << 8 | state << 16
    uint64_t *rowmask = reinterpret_cast<uint64_t *>(lds + ((E * VV + 1) & ~1));   // [j]: bit e*V+i = cell (i, j) of env e visible
    int32_t *topx = reinterpret_cast<int32_t *>(rowmask + V);               // [e] view window origin, rotation count, carried cell
    int32_t *topy = topx + E;
    int32_t *rot = topy + E;
    uint32_t *carried = reinterpret_cast<uint32_t *>(rot + E);
    int32_t *phase = reinterpret_cast<int32_t *>(carried + E);              // [e] (address of the env's image) & 3
    uint8_t *stage = reinterpret_cast<uint8_t *>(phase + E);                // [e][phase + (i*V + j)*3 + ch]: output order, and
                                                                            // 4-byte aligned exactly where the destination is
    const size_t pitch = image_pitch ? (size_t)image_pitch : (size_t)nb;

    const int half = V / 2;
    if (lane < ne) {
        const int n = n0 + lane;
        const int ax = agent_x[n], ay = agent_y[n], dir = agent_dir[n] & 3;
        // get_view_exts (minigrid.py:1262-1293)
        topx[lane] = dir == 0 ? ax : (dir == 2 ? ax - V + 1 : ax - half);
        topy[lane] = dir == 1 ? ay : (dir == 3 ? ay - V + 1 : ay - half);
        rot[lane] = (dir + 1) & 3;                                          // number of rotate_left applications mod 4
        uint32_t c = EMPTY_CELL;
        if (carrying && carrying[(size_t)n * 3] != 0)
            c = carrying[(size_t)n * 3] | ((uint32_t)carrying[(size_t)n * 3 + 1] << 8) |
                ((uint32_t)carrying[(size_t)n * 3 + 2] << 16);
        carried[lane] = c;
        phase[lane] = (int)((uintptr_t)(image + (size_t)n * pitch) & 3u);
    }
    wsync();
    for (int c = lane; c < ne * VV; c += 64) {
        const int e = c / VV, cc = c - e * VV;
        const int j = cc / V, i = cc - j * V;
        const int k = rot[e];
        // rotate_left maps old (a, b) -> new (b, V-1-a); inverted k times: view (i, j) <- slice (si, sj)
        const int si = k == 0 ? i : (k == 1 ? V - 1 - j : (k == 2 ? V - 1 - i : j));
        const int sj = k == 0 ? j : (k == 1 ? i : (k == 2 ? V - 1 - j : V - 1 - i));
        const int x = topx[e] + si, y = topy[e] + sj;
        uint32_t cell = WALL_CELL;                                          // Grid.slice: outside the world -> Wall()
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = (size_t)(n0 + e) * W * H + (size_t)y * W + x;
            const uint32_t t = type[o];
            cell = (t <= T_EMPTY) ? EMPTY_CELL
                                  : (t | ((uint32_t)colour[o] << 8) | ((state ? (uint32_t)state[o] : 0u) << 16));
        }
        cells[c] = cell;
    }
    wsync();

    // ---- process_vis (minigrid.py:795-832), rows from the agent's row (V-1) upwards, all envs of the wave at once
    const uint64_t all = EV >= 64 ? ~0ull : ((1ull << EV) - 1ull);
    if (see_through) {
        if (lane < V) rowmask[lane] = all;
    } else {
        uint64_t seg_first = 0, seed = 0;
        for (int e = 0; e < E; ++e) { seg_first |= 1ull << (e * V); seed |= 1ull << (e * V + half); }   // mask[V//2, V-1] = True
        const uint64_t seg_last = seg_first << (V - 1);
        const int le = lane / V, li = lane - le * V;
        for (int j = V - 1; j >= 0; --j) {
            const bool sb = lane < ne * V && see_behind(cells[le * VV + j * V + li]);
            const uint64_t p = __ballot(sb);
            const uint64_t r = flood_right(seed, (p << 1) & ~seg_first & all);     // first sweep: i = 0 .. V-2
            const uint64_t tl = r & p & ~seg_last;
            const uint64_t r2 = flood_left(r, (p >> 1) & ~seg_last);               // second sweep: i = V-1 .. 1
            const uint64_t tr = r2 & p & ~seg_first;
            if (lane == 0) rowmask[j] = r2;
            seed = tl | (tl << 1) | tr | (tr >> 1);                                // mask[i +- 1, j-1] and mask[i, j-1]
        }
    }
    wsync();

    // ---- encode (minigrid.py:749-772) into output order [i][j][3]; the agent's cell shows the carried object
    for (int c = lane; c < ne * VV; c += 64) {
        const int e = c / VV, cc = c - e * VV;
        const int j = cc / V, i = cc - j * V;
        const bool vis = (rowmask[j] >> (e * V + i)) & 1ull;
        uint32_t cell = cells[c];
        if (i == half && j == V - 1) cell = carried[e];                     // grid.set(*agent_pos, carrying or None)
        if (!vis) cell = 0u;
        uint8_t *s = stage + e * nbp + phase[e] + (i * V + j) * 3;
        s[0] = (uint8_t)cell; s[1] = (uint8_t)(cell >> 8); s[2] = (uint8_t)(cell >> 16);
        if (vis_mask) vis_mask[(size_t)(n0 + e) * VV + i * V + j] = vis ? 1 : 0;
    }
    wsync();
    // ---- copy out: per env, the 4-byte aligned slots that overlap its image; whole slots leave as dwords
    const int slots = (nb + 3) / 4 + 1;
    for (int idx = lane; idx < ne * slots; idx += 64) {
        const int e = idx / slots, d = idx - e * slots;
        uint8_t *dst = image + (size_t)(n0 + e) * pitch;
        const int b0 = 4 * d - phase[e];                                    // first image byte of this slot (may be < 0)
        const uint8_t *s = stage + e * nbp + phase[e];
        if (b0 >= 0 && b0 + 3 < nb) {
            *reinterpret_cast<uint32_t *>(dst + b0) = *reinterpret_cast<const uint32_t *>(s + b0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (b0 + q >= 0 && b0 + q < nb) dst[b0 + q] = s[b0 + q];
        }
    }
}

// Compile-time view sizes: lane = (env slot e, view column i) -- the same layout the ballot masks of process_vis use --
// and the lane keeps its whole column (V cells) in registers: no per-element div / mod.
// ROWS (the shipped mode): the plane bytes arrive row-wise as aligned dwords and are transposed through LDS (see the
// block below); V = 7 occluded 56 -> 41 us, V = 17 see-through 251 -> 143 us on 262 144 random 17x17 worlds.
// !ROWS (MG_VIEW_LOADS=cols, kept for A/B runs): every lane walks its column with one byte load per cell and plane, all
// requested up front.  Counters on that variant (rocprofv3 --pmc, V = 7): a wave lives ~35 k cycles for ~860
// instructions and waits 73 % of that time -- not dependent round trips, as first assumed, but the L1 serving the same
// lines over and over (V byte loads per line, ~130 KB of lines in flight per CU against 32 KB of L1).
// Measured and rejected: G > 1 env groups per wave (G = 2 / 4: 10-60 % slower at every view size, registers cost
// occupancy) and staging each (env, plane) span through LDS with 16-byte loads (89 vs 66 us).
template <int V, int G, bool ROWS>
__global__ __launch_bounds__(64) void mg_gen_obs_cols_kernel(const uint8_t *__restrict__ type, const uint8_t *__restrict__ colour,
                                                             const uint8_t *__restrict__ state, int N, int W, int H,
                                                             const int32_t *__restrict__ agent_x,
                                                             const int32_t *__restrict__ agent_y,
                                                             const int32_t *__restrict__ agent_dir,
                                                             const uint8_t *__restrict__ carrying, int see_through,
                                                             uint8_t *__restrict__ image, int image_pitch,
                                                             uint8_t *__restrict__ vis_mask) {
    constexpr int E = 64 / V, VV = V * V, EV = E * V, nb = VV * 3, nbp = (nb + 6) & ~3, half = V / 2;
    __shared__ __attribute__((aligned(4))) uint8_t stage[G * E * nbp];
    __shared__ int32_t phase[G * E];
    const int lane = threadIdx.x;
    const int le = lane / V, li = lane - le * V, i = li;
    const size_t pitch = image_pitch ? (size_t)image_pitch : (size_t)nb;
    const int nbase = blockIdx.x * (E * G);
    bool act[G];
    int n[G], ax[G], ay[G], dir[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int ng = nbase + g * E + le;
        act[g] = lane < EV && ng < N;
        n[g] = act[g] ? ng : nbase;
        ax[g] = agent_x[n[g]]; ay[g] = agent_y[n[g]]; dir[g] = agent_dir[n[g]] & 3;
    }
    uint32_t cell[G][V], carried[G];
    int ph[G];
    // Every plane byte of every group is requested before any is used: the three planes unconditionally (a colour /
    // state load that waits for the type byte is one more dependent HBM round trip, and its branch keeps the groups
    // from overlapping), out-of-world cells at a clamped, always valid address.
    uint8_t tb[G][V], cb[G][V], sb[G][V];
    bool inw[G][V];
    if constexpr (ROWS) {
        // Row-wise loads: lane (env, r) fetches window row r of its env -- V consecutive bytes per plane -- as ALIGNED
        // dwords, re-aligned in registers (v_alignbyte), so every cache line a row touches is requested by ONE
        // instruction; the byte-per-instruction column walk below asks the L1 for the same lines up to V times, and
        // with ~130 KB of lines in flight per CU they do not survive in the 32 KB L1 between two requests.  Lanes then
        // swap (row, column) roles through a V x V dword tile per env in LDS (odd pitch: conflict-free).
        static_assert(G == 1, "row-wise loads: one env group per wavefront");
        constexpr int ND = ((V - 1) >> 2) + 2;                             // aligned dwords covering V bytes at any alignment
        __shared__ uint32_t win[E * VV];
        const int r = li, lec = le < E ? le : E - 1;
        const int topx = dir[0] == 0 ? ax[0] : (dir[0] == 2 ? ax[0] - V + 1 : ax[0] - half);
        const int topy = dir[0] == 1 ? ay[0] : (dir[0] == 3 ? ay[0] - V + 1 : ay[0] - half);
        const int y = topy + r, yc = min(max(y, 0), H - 1);
        const bool yin = y >= 0 && y < H;
        const intptr_t o0 = (intptr_t)n[0] * W * H + (intptr_t)yc * W + topx;      // window-row byte 0, relative to the plane
        uint32_t rt[ND - 1], rc[ND - 1], rs[ND - 1];
        auto load_row = [&](const uint8_t *plane, uint32_t *out) {
            // dwords are clamped to those that contain bytes of the plane array: an in-world cell's dword always is one
            const intptr_t lo = (intptr_t)((uintptr_t)plane & ~(uintptr_t)3);
            const intptr_t hi = (intptr_t)(((uintptr_t)plane + (size_t)N * W * H - 1) & ~(uintptr_t)3);
            const intptr_t P = (intptr_t)(uintptr_t)plane + o0, A = P & ~(intptr_t)3;
            uint32_t d[ND];
#pragma unroll
            for (int m = 0; m < ND; ++m) {
                intptr_t a = A + 4 * m;
                a = a < lo ? lo : (a > hi ? hi : a);
                d[m] = *reinterpret_cast<const uint32_t *>(a);
            }
            const uint32_t sh = (uint32_t)(P & 3);
#pragma unroll
            for (int m = 0; m < ND - 1; ++m) out[m] = __builtin_amdgcn_alignbyte(d[m + 1], d[m], sh);
        };
        load_row(type, rt);
        load_row(colour, rc);
        if (state) load_row(state, rs);
        else {
#pragma unroll
            for (int m = 0; m < ND - 1; ++m) rs[m] = 0u;
        }
#pragma unroll
        for (int c = 0; c < V; ++c) {
            const int x = topx + c;
            const uint32_t t = (rt[c >> 2] >> (8 * (c & 3))) & 255u, co = (rc[c >> 2] >> (8 * (c & 3))) & 255u,
                           st = (rs[c >> 2] >> (8 * (c & 3))) & 255u;
            const uint32_t cl = (t <= T_EMPTY) ? EMPTY_CELL : (t | (co << 8) | (st << 16));
            if (lane < EV) win[(lec * V + r) * V + c] = (yin && x >= 0 && x < W) ? cl : WALL_CELL;   // Grid.slice: outside -> Wall()
        }
        wsync();
        const int k = (dir[0] + 1) & 3;
        const int sx0 = k == 0 ? i : (k == 1 ? V - 1 : (k == 2 ? V - 1 - i : 0));
        const int sy0 = k == 0 ? 0 : (k == 1 ? i : (k == 2 ? V - 1 : V - 1 - i));
        const int dxj = k == 1 ? -1 : (k == 3 ? 1 : 0), dyj = k == 0 ? 1 : (k == 2 ? -1 : 0);
#pragma unroll
        for (int j = 0; j < V; ++j) cell[0][j] = win[(lec * V + sy0 + j * dyj) * V + sx0 + j * dxj];
    }
#pragma unroll
    for (int g = 0; g < (ROWS ? 0 : G); ++g) {
        // get_view_exts (minigrid.py:1262-1293) and the number of rotate_left applications
        const int topx = dir[g] == 0 ? ax[g] : (dir[g] == 2 ? ax[g] - V + 1 : ax[g] - half);
        const int topy = dir[g] == 1 ? ay[g] : (dir[g] == 3 ? ay[g] - V + 1 : ay[g] - half);
        const int k = (dir[g] + 1) & 3;
        // view (i, j) <- slice (si, sj): rotate_left maps old (a, b) -> new (b, V-1-a), inverted k times.  For this lane i
        // is fixed, so (x, y) walks a straight line in the world as j grows: start + j * step.
        const int sx0 = k == 0 ? i : (k == 1 ? V - 1 : (k == 2 ? V - 1 - i : 0));
        const int sy0 = k == 0 ? 0 : (k == 1 ? i : (k == 2 ? V - 1 : V - 1 - i));
        const int dxj = k == 1 ? -1 : (k == 3 ? 1 : 0), dyj = k == 0 ? 1 : (k == 2 ? -1 : 0);
        const size_t pbase = (size_t)n[g] * W * H;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int x = topx + sx0 + j * dxj, y = topy + sy0 + j * dyj;
            inw[g][j] = act[g] && x >= 0 && x < W && y >= 0 && y < H;
            const int xc = min(max(x, 0), W - 1), yc = min(max(y, 0), H - 1);
            const size_t o = pbase + (size_t)yc * W + xc;
            tb[g][j] = type[o];
            cb[g][j] = colour[o];
            sb[g][j] = state ? state[o] : (uint8_t)0;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int j = 0; j < (ROWS ? 0 : V); ++j) {
            const uint32_t t = tb[g][j];
            const uint32_t c = (t <= T_EMPTY) ? EMPTY_CELL : (t | ((uint32_t)cb[g][j] << 8) | ((uint32_t)sb[g][j] << 16));
            cell[g][j] = inw[g][j] ? c : WALL_CELL;                       // Grid.slice: outside the world -> Wall()
        }
        carried[g] = EMPTY_CELL;
        if (carrying && carrying[(size_t)n[g] * 3] != 0)
            carried[g] = carrying[(size_t)n[g] * 3] | ((uint32_t)carrying[(size_t)n[g] * 3 + 1] << 8) |
                         ((uint32_t)carrying[(size_t)n[g] * 3 + 2] << 16);
        ph[g] = (int)((uintptr_t)(image + (size_t)n[g] * pitch) & 3u);
        if (act[g] && li == 0) phase[g * E + le] = ph[g];
    }
    constexpr uint64_t all = EV >= 64 ? ~0ull : ((1ull << EV) - 1ull);
    uint64_t seg_first = 0, seed0 = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { seg_first |= 1ull << (e * V); seed0 |= 1ull << (e * V + half); }   // mask[V//2, V-1] = True
    const uint64_t seg_last = seg_first << (V - 1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // ---- process_vis (minigrid.py:795-832): rows from the agent's row (V-1) upwards, all envs of the group at once
        uint32_t vis = 0;                                                   // bit j: cell (i, j) of this lane's env is visible
        if (see_through) {
            vis = (1u << V) - 1u;
        } else {
            uint64_t seed = seed0;
#pragma unroll
            for (int j = V - 1; j >= 0; --j) {
                const uint64_t p = __ballot(act[g] && see_behind(cell[g][j]));
                const uint64_t r = flood_right(seed, (p << 1) & ~seg_first & all);     // first sweep: i = 0 .. V-2
                const uint64_t tl = r & p & ~seg_last;
                const uint64_t r2 = flood_left(r, (p >> 1) & ~seg_last);               // second sweep: i = V-1 .. 1
                const uint64_t tr = r2 & p & ~seg_first;
                vis |= (uint32_t)((r2 >> lane) & 1ull) << j;
                seed = tl | (tl << 1) | tr | (tr >> 1);                                // mask[i +- 1, j-1] and mask[i, j-1]
            }
        }
        // ---- encode (minigrid.py:749-772) into output order [i][j][3]; the agent's cell shows the carried object
        if (act[g]) {
            uint8_t *sp = stage + (g * E + le) * nbp + ph[g] + i * V * 3;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                uint32_t c = cell[g][j];
                if (i == half && j == V - 1) c = carried[g];                // grid.set(*agent_pos, carrying or None)
                if (!((vis >> j) & 1u)) c = 0u;
                sp[3 * j] = (uint8_t)c; sp[3 * j + 1] = (uint8_t)(c >> 8); sp[3 * j + 2] = (uint8_t)(c >> 16);
            }
            if (vis_mask) {
#pragma unroll
                for (int j = 0; j < V; ++j) vis_mask[(size_t)n[g] * VV + i * V + j] = (uint8_t)((vis >> j) & 1u);
            }
        }
    }
    wsync();
    // ---- copy out: per env, the 4-byte aligned slots that overlap its image; whole slots leave as dwords
    constexpr int slots = (nb + 3) / 4 + 1;
    const int ne = N - nbase < E * G ? N - nbase : E * G;                   // envs of this wavefront (slot s = g * E + e)
    for (int idx = lane; idx < ne * slots; idx += 64) {
        const int sl = idx / slots, d = idx - sl * slots;
        uint8_t *dst = image + (size_t)(nbase + sl) * pitch;
        const int b0 = 4 * d - phase[sl];                                   // first image byte of this slot (may be < 0)
        const uint8_t *sq = stage + sl * nbp + phase[sl];
        if (b0 >= 0 && b0 + 3 < nb) {
            *reinterpret_cast<uint32_t *>(dst + b0) = *reinterpret_cast<const uint32_t *>(sq + b0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (b0 + q >= 0 && b0 + q < nb) dst[b0 + q] = sq[b0 + q];
        }
    }
}

// _reward() = 1 - 0.9 * (step_count / max_steps) as Python evaluates it: three separately rounded double operations.
// (__dmul_rn / __dsub_rn are plain operators in HIP and would still be contracted into one fma.)
__device__ __forceinline__ double reference_reward(int step_count, int max_steps) {
#pragma clang fp contract(off)
    const double q = (double)step_count / (double)max_steps;
    const double m = 0.9 * q;
    return 1.0 - m;
}

// ---------------------------------------------------------------- MiniGridEnv.step (base class, minigrid.py:1333-1441)
// One thread per env; the world planes are read-only (the reference's reachable actions never edit the grid).
__global__ void mg_step_kernel(const uint8_t *__restrict__ type, const uint8_t *__restrict__ state, int N, int W, int H,
                               const int32_t *__restrict__ action, int32_t *__restrict__ agent_x,
                               int32_t *__restrict__ agent_y, const int32_t *__restrict__ agent_dir,
                               int32_t *__restrict__ step_count, int max_steps, double *__restrict__ reward,
                               uint8_t *__restrict__ terminated, uint8_t *__restrict__ truncated,
                               int32_t *__restrict__ error) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int sc = step_count[n] + 1;                                       // self.step_count += 1 comes first
    step_count[n] = sc;
    const int ax = agent_x[n], ay = agent_y[n], dir = agent_dir[n] & 3, a = action[n];
    const int fx = ax + (dir == 0) - (dir == 2), fy = ay + (dir == 1) - (dir == 3);
    int err = 0, te = 0, tr = 0;
    double r = 0.0;
    if (fx < 0 || fx >= W || fy < 0 || fy >= H) err = 2;                    // fwd_cell = self.grid.get(*fwd_pos): assert
    else if (!(a == 0 || a == 1 || a == 2 || a == 3 || a == 6)) err = 1;   // falls through to `self.actions.forward`
    else {
        const int px = ax - (a == 0) + (a == 1), py = ay - (a == 2) + (a == 3);
        if (px < 0 || px >= W || py < 0 || py >= H) err = 2;
        else {
            const size_t o = (size_t)n * W * H + (size_t)py * W + px;
            const uint32_t t = type[o];
            const bool overlap = t <= T_EMPTY || t == 8u || t == 11u || t == 3u || t == 9u ||
                                 (t == T_DOOR && (state ? state[o] : 0) == 0);
            if (overlap) { agent_x[n] = px; agent_y[n] = py; }
            if (t == 8u) { te = 1; r = reference_reward(sc, max_steps); }                  // _reward(), :1061
            tr = sc >= max_steps;
        }
    }
    reward[n] = r;
    terminated[n] = (uint8_t)te;
    truncated[n] = (uint8_t)tr;
    if (error) error[n] = err;
}

size_t view_lds_bytes(int V, int E) {
    const int VV = V * V;
    return (size_t)((E * VV + 1) & ~1) * 4 + (size_t)V * 8 + (size_t)E * 20 + (size_t)E * ((VV * 3 + 6) & ~3);
}

}  // namespace

extern "C" int mg_gen_obs(const uint8_t *type, const uint8_t *colour, const uint8_t *state, int n_envs, int width,
                          int height, const int32_t *agent_x, const int32_t *agent_y, const int32_t *agent_dir,
                          const uint8_t *carrying, int view_size, int see_through_walls, uint8_t *image, int image_pitch,
                          uint8_t *vis_mask, void *stream) {
    if (!type || !colour || !agent_x || !agent_y || !agent_dir || !image) return TW_E_ARG;
    if (n_envs <= 0 || width <= 0 || height <= 0 || view_size < 1 || view_size > MG_MAX_VIEW) return TW_E_ARG;
    if (image_pitch != 0 && image_pitch < view_size * view_size * 3) return TW_E_ARG;
    const int E = 64 / view_size;                                          // envs per wavefront (2 for V = 31 ... 64 for V = 1)
    const dim3 grid((n_envs + E - 1) / E), block(64);
    const size_t lds = view_lds_bytes(view_size, E);
#define MG_LAUNCH(VT)                                                                                                  \
    hipLaunchKernelGGL(mg_gen_obs_kernel<VT>, grid, block, lds, (hipStream_t)stream, type, colour, state, n_envs, width, \
                       height, agent_x, agent_y, agent_dir, carrying, view_size, see_through_walls ? 1 : 0, image,      \
                       image_pitch, vis_mask)
#define MG_LAUNCH_COLS(VT)                                                                                             \
    do {                                                                                                               \
        constexpr int GG = 1;   /* env groups per wavefront; 2 / 4 measured 10-60 % slower at every view size */        \
        const dim3 grid_g((n_envs + E * GG - 1) / (E * GG));                                                           \
        if (rows)                                                                                                      \
            hipLaunchKernelGGL((mg_gen_obs_cols_kernel<VT, GG, true>), grid_g, block, 0, (hipStream_t)stream, type,    \
                               colour, state, n_envs, width, height, agent_x, agent_y, agent_dir, carrying,            \
                               see_through_walls ? 1 : 0, image, image_pitch, vis_mask);                               \
        else                                                                                                           \
            hipLaunchKernelGGL((mg_gen_obs_cols_kernel<VT, GG, false>), grid_g, block, 0, (hipStream_t)stream, type,   \
                               colour, state, n_envs, width, height, agent_x, agent_y, agent_dir, carrying,            \
                               see_through_walls ? 1 : 0, image, image_pitch, vis_mask);                               \
    } while (0)
    // row-wise aligned-dword loads (the default); MG_VIEW_LOADS=cols keeps the byte-per-cell column walk (A/B diagnostic)
    static const bool cols_env = getenv("MG_VIEW_LOADS") && !strcmp(getenv("MG_VIEW_LOADS"), "cols");
    const bool rows = !cols_env && (size_t)n_envs * width * height >= 64;
    switch (view_size) {                    // the usual odd sizes: lane-per-column kernel with the column in registers
    case 3: MG_LAUNCH_COLS(3); break;
    case 5: MG_LAUNCH_COLS(5); break;
    case 7: MG_LAUNCH_COLS(7); break;
    case 9: MG_LAUNCH_COLS(9); break;
    case 11: MG_LAUNCH_COLS(11); break;
    case 13: MG_LAUNCH_COLS(13); break;
    case 15: MG_LAUNCH_COLS(15); break;
    case 17: MG_LAUNCH_COLS(17); break;
    default: MG_LAUNCH(0); break;
    }
#undef MG_LAUNCH_COLS
#undef MG_LAUNCH
    return hipGetLastError() == hipSuccess ? TW_OK : TW_E_HIP;
}

extern "C" int mg_step(const uint8_t *type, const uint8_t *state, int n_envs, int width, int height,
                       const int32_t *action, int32_t *agent_x, int32_t *agent_y, const int32_t *agent_dir,
                       int32_t *step_count, int max_steps, double *reward, uint8_t *terminated, uint8_t *truncated,
                       int32_t *error, void *stream) {
    if (!type || !action || !agent_x || !agent_y || !agent_dir || !step_count || !reward || !terminated || !truncated)
        return TW_E_ARG;
    if (n_envs <= 0 || width <= 0 || height <= 0 || max_steps <= 0) return TW_E_ARG;
    hipLaunchKernelGGL(mg_step_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, type, state, n_envs,
                       width, height, action, agent_x, agent_y, agent_dir, step_count, max_steps, reward, terminated,
                       truncated, error);
    return hipGetLastError() == hipSuccess ? TW_OK : TW_E_HIP;
}
