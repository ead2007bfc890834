// Single-wave instruction latency / issue probes for the range-decoder chain design (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

#define PROBE(name, body, nper)                                                        \
    __global__ __launch_bounds__(64) void name(uint64_t* out, uint32_t seed) {        \
        uint32_t s = __builtin_amdgcn_readfirstlane(seed), t = s + 1;                  \
        uint32_t v = threadIdx.x + seed, w = v + 1;                                    \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
        asm volatile(REP256(body) : "+s"(s), "+s"(t), "+v"(v), "+v"(w) :: "vcc", "scc", "s40", "s41", "s42", "s43", "m0"); \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + t + v + w; }            \
    }

PROBE(p_salu_dep, "s_add_u32 %0, %0, %1\n", 1)
PROBE(p_salu_indep, "s_add_u32 s40, %0, %1\n", 1)
PROBE(p_salu_mul_dep, "s_mul_i32 %0, %0, %1\n", 1)
PROBE(p_valu_dep, "v_add_u32 %2, %2, %3\n", 1)
PROBE(p_valu_indep, "v_add_u32 %3, %2, %2\n", 1)
PROBE(p_mad64_dep, "v_mad_u64_u32 v[10:11], vcc, %2, %3, v[10:11]\n", 1)
PROBE(p_readlane_salu, "v_readlane_b32 s40, %2, %0\n s_and_b32 %0, s40, 63\n", 2)
PROBE(p_salu_valu_readlane, "v_add_u32 %2, %0, %2\n v_readlane_b32 %0, %2, 3\n", 2)
PROBE(p_cmp_bcnt, "v_cmp_ge_u32 vcc, %2, %3\n s_bcnt1_i32_b64 s40, vcc\n v_add_u32 %2, s40, %2\n", 3)
PROBE(p_cmp_ff1_readlane, "v_cmp_ge_u32 vcc, %2, %3\n s_ff1_i32_b64 s40, vcc\n v_readlane_b32 s41, %3, s40\n v_add_u32 %2, s41, %2\n", 4)
PROBE(p_branch_nt, "s_cmp_eq_u32 %0, 0x12345\n s_cbranch_scc1 99f\n s_add_u32 %0, %0, 1\n99:\n", 3)
PROBE(p_dsread_dep, "ds_read_b32 %2, %2\n s_waitcnt lgkmcnt(0)\n v_and_b32 %2, 0xffc, %2\n", 3)
PROBE(p_mix_indep, "s_add_u32 s40, %0, %1\n v_add_u32 %3, %2, %2\n", 2)

typedef void (*kern_t)(uint64_t*, uint32_t);
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 64));
    struct { const char* n; kern_t k; int per; } P[] = {
        {"SALU dependent add", p_salu_dep, 1}, {"SALU independent add", p_salu_indep, 1}, {"SALU dependent mul", p_salu_mul_dep, 1},
        {"VALU dependent add", p_valu_dep, 1}, {"VALU independent add", p_valu_indep, 1}, {"v_mad_u64_u32 dependent", p_mad64_dep, 1},
        {"readlane->SALU->readlane (2 instr)", p_readlane_salu, 2}, {"VALU(sgpr)->readlane->VALU (2 instr)", p_salu_valu_readlane, 2},
        {"v_cmp->s_bcnt->v_add (3 instr)", p_cmp_bcnt, 3}, {"v_cmp->s_ff1->v_readlane->v_add (4 instr)", p_cmp_ff1_readlane, 4},
        {"s_cmp+branch not taken+s_add (3 instr)", p_branch_nt, 3}, {"ds_read dependent chain (3 instr)", p_dsread_dep, 3},
        {"independent SALU+VALU pair (2 instr)", p_mix_indep, 2}};
    for (auto& p : P) {
        uint64_t h[2];
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, d, 5u); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%-45s %7.2f memtime ticks / group, %6.2f / instr\n", p.n, h[0] / 256.0, h[0] / 256.0 / p.per);
    }
    // calibrate s_memtime against wall: long dependent SALU chain
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(p_salu_dep, dim3(1), dim3(64), 0, 0, d, 5u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("200 launches of 256 dependent s_add: %.3f ms total\n", ms);
    return 0;
}
