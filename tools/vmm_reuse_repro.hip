// HIP-only reproducer for the slab incident of DESIGN.md (engine slab): a virtual range released with hipMemAddressFree
// and handed out again by hipMemAddressReserve -- does a kernel writing through the new mapping lose rows?
//   hipcc --offload-arch=gfx950 -O2 -o tools/vmm_reuse_repro tools/vmm_reuse_repro.hip
//   tools/vmm_reuse_repro <sync_before_unmap 0|1> <address_free 0|1> [rounds=3]
// Prints one line per round: same_va, rows lost as seen by a device-side check and by a host copy.  Run once per mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
constexpr size_t ROW = 2048, CHUNK = (size_t)2 << 20, BYTES = (size_t)1 << 30, ROWS = BYTES / ROW;

__global__ void write_rows(uint4 *p, unsigned stamp) {          // one wave per row, 2 x dwordx4 per lane like the engine
    const size_t row = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned lane = threadIdx.x & 63;
    uint4 v = make_uint4(stamp, (unsigned)row, lane, ~stamp);
    p[row * (ROW / 16) + lane] = v;
    p[row * (ROW / 16) + 64 + lane] = v;
}
__global__ void count_bad(const uint4 *p, unsigned stamp, unsigned long long *bad) {
    const size_t row = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const unsigned lane = threadIdx.x & 63;
    const uint4 a = p[row * (ROW / 16) + lane], b = p[row * (ROW / 16) + 64 + lane];
    const bool ok = a.x == stamp && a.y == (unsigned)row && a.z == lane && b.x == stamp && b.y == (unsigned)row;
    if (__any(!ok) && lane == 0) atomicAdd(bad, 1ull);
}
struct Slab { void *base = nullptr; std::vector<hipMemGenericAllocationHandle_t> h; };

static void slab_map(Slab &s, hipMemAllocationProp &prop, size_t gran) {
    CK(hipMemAddressReserve(&s.base, BYTES, gran, nullptr, 0));
    for (size_t i = 0; i < BYTES / CHUNK; ++i) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, CHUNK, &prop, 0));
        CK(hipMemMap((char *)s.base + i * CHUNK, CHUNK, 0, h, 0));
        s.h.push_back(h);
    }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(s.base, BYTES, &acc, 1));
}
static void slab_unmap(Slab &s, bool address_free) {
    for (size_t i = 0; i < s.h.size(); ++i) CK(hipMemUnmap((char *)s.base + i * CHUNK, CHUNK));
    for (auto h : s.h) CK(hipMemRelease(h));
    s.h.clear();
    if (address_free) CK(hipMemAddressFree(s.base, BYTES));
}
int main(int argc, char **argv) {
    const bool sync = argc > 1 ? atoi(argv[1]) != 0 : true, afree = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int rounds = argc > 3 ? atoi(argv[3]) : 3;
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran < CHUNK) gran = CHUNK;
    unsigned long long *bad; CK(hipMalloc(&bad, 8));
    void *other; CK(hipMalloc(&other, BYTES));
    std::vector<unsigned> host(BYTES / 4);
    void *prev = nullptr; int total_lost = 0;
    for (int r = 0; r < rounds; ++r) {
        Slab s; slab_map(s, prop, gran);
        const unsigned stamp = 0xA5000000u + r;
        CK(hipMemsetAsync(s.base, 0x77, BYTES, 0));
        write_rows<<<ROWS / 4, 256>>>((uint4 *)s.base, stamp);
        CK(hipMemsetAsync(bad, 0, 8, 0));
        count_bad<<<ROWS / 4, 256>>>((const uint4 *)s.base, stamp, bad);
        unsigned long long dev_bad = 0; CK(hipMemcpy(&dev_bad, bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(host.data(), s.base, BYTES, hipMemcpyDeviceToHost));
        size_t host_bad = 0;
        for (size_t row = 0; row < ROWS; ++row) host_bad += host[row * (ROW / 4)] != stamp || host[row * (ROW / 4) + 1] != (unsigned)row;
        printf("round %d sync=%d address_free=%d same_va_as_previous=%d rows_lost_device=%llu rows_lost_host=%zu of %zu\n",
               r, (int)sync, (int)afree, (int)(s.base == prev), dev_bad, host_bad, ROWS);
        total_lost += (int)(dev_bad + host_bad);
        // the step under test: tear the mapping down behind a drained device (sync) or while the device is still busy with
        // kernels that touch ANOTHER allocation (never the range being unmapped: that would be a page fault by design)
        for (int k = 0; k < 4; ++k) write_rows<<<ROWS / 4, 256>>>((uint4 *)other, stamp ^ 0xFFFFu);
        if (sync) CK(hipDeviceSynchronize());
        prev = s.base;
        slab_unmap(s, afree);
    }
    CK(hipDeviceSynchronize());
    printf("RESULT sync=%d address_free=%d rows_lost_total=%d\n", (int)sync, (int)afree, total_lost);
    return 0;
}
