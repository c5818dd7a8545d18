// minigrid_view.hip -- general MiniGrid agent view for MI355X (gfx950), C ABI in include/minigrid_view.h.
//
// gen_obs_grid + process_vis + encode of gym_minigrid/minigrid.py:1443-1496 / :795-832 / :749-772 for N envs
// with arbitrary W x H worlds, agent direction, view size, carried object and occlusion.
//
// One wavefront per env.  The V x V window is gathered straight into its rotated position (the dir+1
// rotate_left calls of the reference are a closed-form index map), one cell per lane-iteration, into LDS.
// process_vis is a row-sequential flood; within a row both of its sweeps are carry chains, so a whole row is
// resolved with 64-bit mask arithmetic on the scalar unit: the row's "see behind" bits come from one ballot,
// the left-to-right sweep is r[i] = m[i] | (r[i-1] & p[i-1]) (Kogge-Stone over the mask), the right-to-left
// sweep its mirror image, and the seeds of the next row are shifts of (reached & see-behind).
// The encoded image is staged in LDS in output order and leaves as coalesced dwords.

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// r[i] = m[i] | (r[i-1] & p[i-1]): bits of m flood to the right through set bits of p
__device__ __forceinline__ uint64_t flood_right(uint64_t m, uint64_t p) {
    uint64_t g = m, q = p << 1;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) { g |= (g << d) & q; q &= q << d; }
    return g;
}

__device__ __forceinline__ uint64_t flood_left(uint64_t m, uint64_t p) {
    uint64_t g = m, q = p >> 1;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) { g |= (g >> d) & q; q &= q >> d; }
    return g;
}

__global__ __launch_bounds__(64) void mg_gen_obs_kernel(const uint8_t *__restrict__ type, const uint8_t *__restrict__ colour,
                                                        const uint8_t *__restrict__ state, int N, int W, int H,
                                                        const int32_t *__restrict__ agent_x,
                                                        const int32_t *__restrict__ agent_y,
                                                        const int32_t *__restrict__ agent_dir,
                                                        const uint8_t *__restrict__ carrying, int V, int see_through,
                                                        uint8_t *__restrict__ image, int image_pitch,
                                                        uint8_t *__restrict__ vis_mask) {
    extern __shared__ uint32_t lds[];
    const int n = blockIdx.x, lane = threadIdx.x;
    if (n >= N) return;
    const int VV = V * V;
    uint32_t *cells = lds;                                                  // [j][i], packed type | colour << 8 | state << 16
    uint64_t *rowmask = reinterpret_cast<uint64_t *>(lds + ((VV + 1) & ~1)); // [j]: bit i = cell (i, j) is visible
    uint8_t *stage = reinterpret_cast<uint8_t *>(rowmask + V);              // image bytes in output order (+ alignment phase)

    const int ax = agent_x[n], ay = agent_y[n], dir = agent_dir[n] & 3;
    // get_view_exts (minigrid.py:1262-1293)
    const int half = V / 2;
    const int topX = dir == 0 ? ax : (dir == 2 ? ax - V + 1 : ax - half);
    const int topY = dir == 1 ? ay : (dir == 3 ? ay - V + 1 : ay - half);
    const size_t plane = (size_t)n * W * H;
    const int k = (dir + 1) & 3;                                            // number of rotate_left applications mod 4
    for (int c = lane; c < VV; c += 64) {
        const int j = c / V, i = c - j * V;
        // rotate_left maps old (a, b) -> new (b, V-1-a); inverted k times: view (i, j) <- slice (si, sj)
        const int si = k == 0 ? i : (k == 1 ? V - 1 - j : (k == 2 ? V - 1 - i : j));
        const int sj = k == 0 ? j : (k == 1 ? i : (k == 2 ? V - 1 - j : V - 1 - i));
        const int x = topX + si, y = topY + sj;
        uint32_t cell = WALL_CELL;                                          // Grid.slice: outside the world -> Wall()
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const size_t o = plane + (size_t)y * W + x;
            const uint32_t t = type[o];
            cell = (t <= T_EMPTY) ? EMPTY_CELL
                                  : (t | ((uint32_t)colour[o] << 8) | ((state ? (uint32_t)state[o] : 0u) << 16));
        }
        cells[c] = cell;
    }
    wsync();

    // ---- process_vis (minigrid.py:795-832), rows from the agent's row (V-1) upwards
    const uint64_t all = V >= 64 ? ~0ull : ((1ull << V) - 1ull);
    if (see_through) {
        if (lane < V) rowmask[lane] = all;
    } else {
        uint64_t seed = 1ull << half;                                       // mask[agent_pos] = True, agent_pos = (V//2, V-1)
        for (int j = V - 1; j >= 0; --j) {
            const bool sb = lane < V && see_behind(cells[j * V + lane]);
            const uint64_t p = __ballot(sb);
            const uint64_t r = flood_right(seed, p) & all;                  // first sweep: i = 0 .. V-2
            const uint64_t tl = r & p & (all >> 1);
            const uint64_t r2 = flood_left(r, p) & all;                     // second sweep: i = V-1 .. 1
            const uint64_t tr = r2 & p & ~1ull;
            if (lane == 0) rowmask[j] = r2;
            seed = (tl | (tl << 1) | tr | (tr >> 1)) & all;                 // mask[i +- 1, j-1] and mask[i, j-1]
        }
    }
    wsync();

    // ---- encode (minigrid.py:749-772) into output order [i][j][3]; the agent's cell shows the carried object
    const size_t pitch = image_pitch ? (size_t)image_pitch : (size_t)VV * 3;
    uint8_t *dst = image + (size_t)n * pitch;
    const unsigned phase = (unsigned)((uintptr_t)dst & 3u);
    uint32_t carried = EMPTY_CELL;
    if (carrying && carrying[(size_t)n * 3] != 0)
        carried = carrying[(size_t)n * 3] | ((uint32_t)carrying[(size_t)n * 3 + 1] << 8) |
                  ((uint32_t)carrying[(size_t)n * 3 + 2] << 16);
    for (int c = lane; c < VV; c += 64) {
        const int j = c / V, i = c - j * V;
        const bool vis = (rowmask[j] >> i) & 1ull;
        uint32_t cell = cells[c];
        if (i == half && j == V - 1) cell = carried;                        // grid.set(*agent_pos, carrying or None)
        if (!vis) cell = 0u;
        uint8_t *s = stage + phase + (size_t)(i * V + j) * 3;
        s[0] = (uint8_t)cell; s[1] = (uint8_t)(cell >> 8); s[2] = (uint8_t)(cell >> 16);
        if (vis_mask) vis_mask[(size_t)n * VV + i * V + j] = vis ? 1 : 0;
    }
    wsync();
    // stage and dst share their alignment phase: head bytes, aligned dwords, tail bytes
    const int nb = VV * 3;
    const int first = (int)((4u - phase) & 3u) < nb ? (int)((4u - phase) & 3u) : nb;
    const int nmid = (nb - first) >> 2, tail = (nb - first) & 3;
    if (lane < first) dst[lane] = stage[phase + lane];
    {
        uint32_t *gd = reinterpret_cast<uint32_t *>(dst + first);
        const uint32_t *sd = reinterpret_cast<const uint32_t *>(stage + phase + first);
        for (int d = lane; d < nmid; d += 64) gd[d] = sd[d];
    }
    if (lane < tail) dst[first + nmid * 4 + lane] = stage[phase + first + nmid * 4 + lane];
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

size_t view_lds_bytes(int V) {
    const int VV = V * V;
    return (size_t)((VV + 1) & ~1) * 4 + (size_t)V * 8 + (size_t)((VV * 3 + 8 + 3) & ~3);
}

}  // namespace

extern "C" int mg_gen_obs(const uint8_t *type, const uint8_t *colour, const uint8_t *state, int n_envs, int width,
                          int height, const int32_t *agent_x, const int32_t *agent_y, const int32_t *agent_dir,
                          const uint8_t *carrying, int view_size, int see_through_walls, uint8_t *image, int image_pitch,
                          uint8_t *vis_mask, void *stream) {
    if (!type || !colour || !agent_x || !agent_y || !agent_dir || !image) return TW_E_ARG;
    if (n_envs <= 0 || width <= 0 || height <= 0 || view_size < 1 || view_size > MG_MAX_VIEW) return TW_E_ARG;
    if (image_pitch != 0 && image_pitch < view_size * view_size * 3) return TW_E_ARG;
    hipLaunchKernelGGL(mg_gen_obs_kernel, dim3(n_envs), dim3(64), view_lds_bytes(view_size), (hipStream_t)stream, type,
                       colour, state, n_envs, width, height, agent_x, agent_y, agent_dir, carrying, view_size,
                       see_through_walls ? 1 : 0, image, image_pitch, vis_mask);
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
