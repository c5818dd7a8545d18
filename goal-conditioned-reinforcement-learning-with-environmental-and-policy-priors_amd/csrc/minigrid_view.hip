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
