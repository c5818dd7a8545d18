// What does ONE wavefront pay per instruction on gfx950?  (the pipelined kernel's logic wave is a lone dependent chain)
//   hipcc --offload-arch=gfx950 -O2 -o tools/issue_probe tools/issue_probe.hip && tools/issue_probe
// Each test = 512 copies of a small instruction group between two s_memtime reads, one wave, repeated; prints ticks per
// instruction (s_memtime ticks; the logic wave's stamps are in the same unit).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define REP 512
#define TEST(name, n_ins, body)                                                                                  \
    __global__ void name(unsigned long long *out, unsigned seed) {                                                \
        unsigned v0 = seed + threadIdx.x, v1 = seed * 3u, v2 = 5u, v3 = 7u, v4 = 11u;                             \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                               \
        asm volatile(".rept 512\n\t" body "\n\t.endr\n9:" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4) :: "vcc", "s20", "s21", "s22", "s23", "memory"); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                               \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0 + v1 + v2 + v3 + v4; }                               \
    }                                                                                                             \
    static const int name##_n = n_ins;
TEST(dep_add, 1, "v_add_u32 %0, %0, %1")
TEST(indep_add, 4, "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4")
TEST(cmp_cnd_vcc, 2, "v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %3, vcc")
TEST(cmp_nop_cnd, 3, "v_cmp_eq_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %2, %3, vcc")
TEST(cmp_fill_cnd, 4, "v_cmp_eq_u32 vcc, %0, %1\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\tv_cndmask_b32 %0, %2, %3, vcc")
TEST(cmp_sgpr_cnd, 2, "v_cmp_eq_u32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %2, %3, s[20:21]")
TEST(smov_vbfe, 2, "s_mov_b32 s20, 0x19019\n\tv_bfe_u32 %0, s20, %0, 2")
TEST(salu_chain, 1, "s_add_u32 s20, s20, 1")
TEST(salu_valu_mix, 2, "s_add_u32 s20, s20, 1\n\tv_add_u32 %0, %0, %1")
TEST(bitop3, 1, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
TEST(lshl_or, 1, "v_lshl_or_b32 %0, %1, 3, %0")
TEST(saveexec, 3, "v_cmp_eq_u32 vcc, %0, %1\n\ts_and_saveexec_b64 s[22:23], vcc\n\ts_or_b64 exec, exec, s[22:23]")
TEST(snop0, 1, "s_nop 0")
TEST(cmp_vccnz, 2, "v_cmp_gt_u32 vcc, %0, %0\n\ts_cbranch_vccnz 9f")
TEST(cmp_scc, 3, "v_cmp_gt_u32 vcc, %0, %0\n\ts_cmp_lg_u64 vcc, 0\n\ts_cbranch_scc1 9f")
TEST(cmp_execnz, 4, "v_cmp_gt_u32 vcc, %0, %0\n\ts_and_saveexec_b64 s[22:23], vcc\n\ts_cbranch_execnz 9f\n\ts_or_b64 exec, exec, s[22:23]")
TEST(dep_add_prio, 1, "v_add_u32 %0, %0, %1")
__global__ void branch_taken(unsigned long long *out, unsigned seed) {
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile(".rept 512\n\ts_branch 1f\n\ts_nop 0\n\ts_nop 0\n1:\n\t.endr" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = seed; }
}
static const int branch_taken_n = 1;
#define RUN(name) do { name<<<1, 64>>>(d, 1u); CK(hipDeviceSynchronize()); name<<<1, 64>>>(d, 2u); CK(hipDeviceSynchronize());      \
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));                                                                     \
        printf("%-14s %6.2f ticks per instruction (%d per group, %llu ticks for %d groups)\n", #name,                       \
               (double)h[0] / (REP * name##_n), name##_n, h[0], REP); } while (0)
int main() {
    unsigned long long *d, h[2];
    CK(hipMalloc(&d, 16));
    RUN(dep_add); RUN(indep_add); RUN(cmp_cnd_vcc); RUN(cmp_nop_cnd); RUN(cmp_fill_cnd); RUN(cmp_sgpr_cnd); RUN(smov_vbfe);
    RUN(salu_chain); RUN(salu_valu_mix); RUN(bitop3); RUN(lshl_or); RUN(saveexec); RUN(snop0); RUN(cmp_vccnz); RUN(cmp_scc); RUN(cmp_execnz); RUN(branch_taken);
    return 0;
}
