// store_pattern_probe.hip -- write-only microbenchmark for the emission store pattern of tw_pipe_kernel.
// Diagnostic tool: how fast can 256 workgroups x 16 waves write time-major [T][N][row]
// outputs when every wave emits G consecutive rows per task and tasks are handed out in (step, group) order?
//
//   hipcc -O3 --offload-arch=gfx950 -o store_pattern_probe tools/store_pattern_probe.hip && ./store_pattern_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// mode 0: linear fill (grid-stride 16 B per lane)
__global__ void fill_kernel(uint4 *dst, size_t n16) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

// mode 1: pipe-like.  Workgroup b owns envs [PG*b, PG*b + PG); its `waves` waves pull tasks (t, g) from a shared
// counter in step-major order; a task writes G consecutive rows of `row` bytes of each of the two streams.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void pipe_kernel(uint8_t *obs, uint8_t *mat, int T, int N, int PG, int G, int row_obs,
                                                          int row_mat, int pace, int block_major = 0, int split = 0) {
    __shared__ int ctr;
    if (threadIdx.x == 0) ctr = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n0 = blockIdx.x * PG;
    const int gps = (PG / G) << split;                   // split: a task writes ONE stream of its rows (twice the tasks)
    const int ntask = T * gps;
    const uint4 v = make_uint4(lane, blockIdx.x, 3, 4);
    while (true) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&ctr, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= ntask) break;
        const int t = k / gps;
        int g = k - t * gps;
        const int which = split ? (g & 1) : 2;                   // 0 obs only, 1 matrix only, 2 both
        g >>= split;
        // time-major [T][N] rows, or workgroup-major [N/PG][T][PG] rows (every workgroup streams through its own region)
        const size_t r0 = block_major ? ((size_t)blockIdx.x * T + t) * PG + g * G : (size_t)t * N + n0 + g * G;
        if (row_obs && which != 1) {
            uint8_t *d = obs + r0 * row_obs;
            for (int off = 16 * lane; off < G * row_obs; off += 1024) *reinterpret_cast<uint4 *>(d + off) = v;
        }
        if (row_mat && which != 0) {
            uint8_t *d = mat + r0 * row_mat;
            for (int off = 16 * lane; off < G * row_mat; off += 1024) *reinterpret_cast<uint4 *>(d + off) = v;
        }
        for (int p = 0; p < pace; ++p) __builtin_amdgcn_s_sleep(8);          // emulate compute between tasks
    }
}

int main() {
    const int T = 128, N = 4096;
    const int row_obs = 880, row_mat = 1168;
    uint8_t *obs, *mat;
    CK(hipMalloc(&obs, (size_t)T * N * 1024 + 4096));
    CK(hipMalloc(&mat, (size_t)T * N * 2048 + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, size_t bytes, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.1f us  %7.1f GB/s\n", name, ms * 100.f, bytes / (ms / 10 * 1e-3) / 1e9);
        fflush(stdout);
    };
    const size_t both = (size_t)T * N * (row_obs + row_mat);
    timeit("fill 1.07 GB linear", both, [&] { hipLaunchKernelGGL(fill_kernel, dim3(256 * 8), dim3(256), 0, 0, (uint4 *)mat, both / 16); });
    char name[128];
    for (int PG : {16, 32, 64}) {
        for (int G : {1, 2, 4, 8, 16}) {
            if (G > PG) continue;
            for (int pace : {0, 4}) {
                snprintf(name, sizeof name, "pipe 16 waves PG=%d G=%d pace=%d obs+mat", PG, G, pace);
                timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / PG), dim3(1024), 0, 0, obs, mat, T, N, PG, G, row_obs, row_mat, pace); });
            }
        }
    }
    for (int G : {1, 8}) {
        snprintf(name, sizeof name, "pipe 16 waves PG=16 G=%d mat only (1168 B rows)", G);
        timeit(name, (size_t)T * N * row_mat, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, G, 0, row_mat, 0); });
        snprintf(name, sizeof name, "pipe 16 waves PG=16 G=%d mat only (2048 B rows)", G);
        timeit(name, (size_t)T * N * 2048, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, G, 0, 2048, 0); });
    }
    // placement of the two streams relative to each other inside one slab (DRAM channel / bank aliasing)
    {
        uint8_t *slab;
        const size_t mat_bytes = (size_t)T * N * row_mat, obs_bytes = (size_t)T * N * row_obs;
        CK(hipMalloc(&slab, mat_bytes + obs_bytes + (64u << 20)));
        for (size_t delta : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144,
                             (size_t)1048576, (size_t)(3u << 20), (size_t)(16u << 20), (size_t)(33u << 20)}) {
            for (int G : {1, 2}) {
                snprintf(name, sizeof name, "slab: obs = mat_end + %zu B, PG=16 G=%d pace=4", delta, G);
                uint8_t *m = slab, *o = slab + mat_bytes + delta;
                timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, o, m, T, N, 16, G, row_obs, row_mat, 4); });
            }
        }
        if (getenv("PROBE_DELTA_SWEEP")) {
            hipFree(slab);
            CK(hipMalloc(&slab, mat_bytes + obs_bytes + (272u << 20)));
            for (size_t mb = 0; mb <= 256; mb += 2) {
                snprintf(name, sizeof name, "sweep: obs = mat_end + %zu MiB, G=1 pace=4", mb);
                uint8_t *m = slab, *o = slab + mat_bytes + (mb << 20);
                timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, o, m, T, N, 16, 1, row_obs, row_mat, 4); });
            }
            for (size_t kb = 0; kb <= 2048; kb += 64) {
                snprintf(name, sizeof name, "sweep: obs = mat_end + %zu KiB, G=1 pace=4", kb);
                uint8_t *m = slab, *o = slab + mat_bytes + (kb << 10);
                timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, o, m, T, N, 16, 1, row_obs, row_mat, 4); });
            }
        }
        for (int rep = 0; rep < 4; ++rep) {
            snprintf(name, sizeof name, "repeat separate allocations PG=16 G=1 pace=4 (#%d)", rep);
            timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, 4); });
        }
        hipFree(slab);
    }
    // row pitches: do line-aligned rows (896 = 7 x 128, 1280 = 10 x 128) pay for the extra bytes?
    for (int pace : {0, 4}) {
        const int pitches[][2] = {{880, 1168}, {896, 1168}, {896, 1280}, {1024, 1280}, {880, 1280}};
        for (auto &pp : pitches) {
            snprintf(name, sizeof name, "pitch obs %d mat %d, PG=16 G=1 pace=%d (time only)", pp[0], pp[1], pace);
            timeit(name, (size_t)T * N * (pp[0] + pp[1]), [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, pp[0], pp[1], pace); });
        }
    }
    for (int pace : {0, 4}) {
        for (int G : {1, 2, 8, 16}) {
            snprintf(name, sizeof name, "workgroup-major layout [N/16][T][16], G=%d pace=%d", G, pace);
            timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, G, row_obs, row_mat, pace, 1); });
        }
        snprintf(name, sizeof name, "time-major layout [T][N], G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace, 0); });
    }
    for (int pace : {0, 2, 4}) {
        snprintf(name, sizeof name, "one stream per task (split), PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace, 0, 1); });
        snprintf(name, sizeof name, "both streams per task,          PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace, 0, 0); });
    }
    // fewer writer waves per workgroup (256 workgroups x PG=16, one row per task)
    for (int pace : {0, 4, 16}) {
        snprintf(name, sizeof name, "writers: 4 waves / workgroup, PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<4>, dim3(N / 16), dim3(256), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace); });
        snprintf(name, sizeof name, "writers: 8 waves / workgroup, PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<8>, dim3(N / 16), dim3(512), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace); });
        snprintf(name, sizeof name, "writers: 12 waves / workgroup, PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<12>, dim3(N / 16), dim3(768), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace); });
        snprintf(name, sizeof name, "writers: 16 waves / workgroup, PG=16 G=1 pace=%d", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 16), dim3(1024), 0, 0, obs, mat, T, N, 16, 1, row_obs, row_mat, pace); });
        snprintf(name, sizeof name, "writers: 16 waves / workgroup, PG=32 G=1 pace=%d (128 workgroups)", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<16>, dim3(N / 32), dim3(1024), 0, 0, obs, mat, T, N, 32, 1, row_obs, row_mat, pace); });
        snprintf(name, sizeof name, "writers: 8 waves / workgroup, PG=32 G=1 pace=%d (128 workgroups)", pace);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<8>, dim3(N / 32), dim3(512), 0, 0, obs, mat, T, N, 32, 1, row_obs, row_mat, pace); });
    }
    for (int G : {1, 8}) {
        snprintf(name, sizeof name, "pipe 8 waves x 512 blocks PG=8 G=%d obs+mat", G);
        timeit(name, both, [&] { hipLaunchKernelGGL(pipe_kernel<8>, dim3(N / 8), dim3(512), 0, 0, obs, mat, T, N, 8, G > 8 ? 8 : G, row_obs, row_mat, 0); });
    }
    hipFree(obs); hipFree(mat);
    return 0;
}
