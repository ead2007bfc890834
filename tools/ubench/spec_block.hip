// How fast can the range-decoder recurrence run when the symbol is PREDICTED (the mode of its window) instead of searched?
// State (dist, range) lives in VGPRs (lane-uniform), the predicted symbol's (L, P) come from a lane of a vector register
// (v_readlane -> SGPR operand, independent of the recurrence), validity is a compare whose result is only looked at once
// per block of 4 symbols.  Compared with the production symbol loop (tools/ubench/dloop.hip: 164 ticks / symbol).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/spec_block.hip -o tools/ubench/spec_block
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one symbol: scale = range >> 24; nr = scale * P; off = scale * L; nd = dist - off; ok = (nd < nr) (a renormalisation shows as nr >> 32 == 0, checked with the block)
#define SYM(LREG, PREG, MASK)                                                   \
    "v_lshrrev_b64 v[10:11], 24, v[6:7]\n\t"                                    \
    "v_mad_u64_u32 v[12:13], s[20:21], v10, " PREG ", 0\n\t"                    \
    "v_mad_u32_u24 v13, v11, " PREG ", v13\n\t"                                 \
    "v_mad_u64_u32 v[14:15], s[20:21], v10, " LREG ", 0\n\t"                    \
    "v_mad_u32_u24 v15, v11, " LREG ", v15\n\t"                                 \
    "v_sub_co_u32 v4, vcc, v4, v14\n\t"                                         \
    "v_subb_co_u32 v5, vcc, v5, v15, vcc\n\t"                                   \
    "v_cmp_lt_u64 " MASK ", v[4:5], v[12:13]\n\t"                               \
    "v_mov_b32 v6, v12\n\t"                                                     \
    "v_mov_b32 v7, v13\n\t"

__global__ __launch_bounds__(64) void spec_kernel(uint64_t* out, uint32_t l0, uint32_t p0, int blocks) {
    const int lane = threadIdx.x;
    uint32_t vl = l0 + lane, vp = p0 - lane;       // (L, P) of the batch's 16 predicted symbols in lanes 0..15
    uint32_t dlo = 0x89abcdefu, dhi = 0x01234567u, rlo = 0x76543210u, rhi = 0xfedcba98u;
    uint32_t okacc = 0xffffffffu;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < blocks; ++b) {
        uint32_t ok;
        asm volatile(
            "v_mov_b32 v4, %[dlo]\n\t v_mov_b32 v5, %[dhi]\n\t v_mov_b32 v6, %[rlo]\n\t v_mov_b32 v7, %[rhi]\n\t"
            "v_readlane_b32 s24, %[vl], 0\n\t v_readlane_b32 s25, %[vp], 0\n\t"
            "v_readlane_b32 s26, %[vl], 1\n\t v_readlane_b32 s27, %[vp], 1\n\t"
            "v_readlane_b32 s28, %[vl], 2\n\t v_readlane_b32 s29, %[vp], 2\n\t"
            "v_readlane_b32 s30, %[vl], 3\n\t v_readlane_b32 s31, %[vp], 3\n\t"
            SYM("s24", "s25", "s[32:33]")
            SYM("s26", "s27", "s[34:35]")
            SYM("s28", "s29", "s[36:37]")
            SYM("s30", "s31", "s[38:39]")
            "s_and_b64 s[32:33], s[32:33], s[34:35]\n\t"
            "s_and_b64 s[36:37], s[36:37], s[38:39]\n\t"
            "s_and_b64 s[32:33], s[32:33], s[36:37]\n\t"
            "s_mov_b32 %[ok], s32\n\t"
            "v_readfirstlane_b32 %[dlo], v4\n\t v_readfirstlane_b32 %[dhi], v5\n\t"
            "v_readfirstlane_b32 %[rlo], v6\n\t v_readfirstlane_b32 %[rhi], v7\n\t"
            : [dlo] "+s"(dlo), [dhi] "+s"(dhi), [rlo] "+s"(rlo), [rhi] "+s"(rhi), [ok] "=s"(ok)
            : [vl] "v"(vl), [vp] "v"(vp)
            : "vcc", "v4", "v5", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "s20", "s21", "s24", "s25", "s26", "s27", "s28", "s29", "s30",
              "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39");
        okacc &= ok;
        rhi |= 0xf0000000u;               // keep the recurrence alive for the timing loop
        if (okacc == 0x12345u) break;     // a (never taken) branch per block, like the real hit / miss decision
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = dlo + dhi + rlo + rhi + okacc; }
}

// the same with the state left in VGPRs between blocks (no readfirstlane per block): what a long hit run costs
__global__ __launch_bounds__(64) void spec_kernel_vstate(uint64_t* out, uint32_t l0, uint32_t p0, int blocks) {
    const int lane = threadIdx.x;
    uint32_t vl = l0 + lane, vp = p0 - lane;
    uint32_t d0 = 0x89abcdefu, d1 = 0x01234567u, r0 = 0x76543210u, r1 = 0xfedcba98u;
    uint32_t okacc = 0xffffffffu;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < blocks; ++b) {
        uint32_t ok;
        asm volatile(
            "v_mov_b32 v4, %[d0]\n\t v_mov_b32 v5, %[d1]\n\t v_mov_b32 v6, %[r0]\n\t v_mov_b32 v7, %[r1]\n\t"
            "v_readlane_b32 s24, %[vl], 0\n\t v_readlane_b32 s25, %[vp], 0\n\t"
            "v_readlane_b32 s26, %[vl], 1\n\t v_readlane_b32 s27, %[vp], 1\n\t"
            "v_readlane_b32 s28, %[vl], 2\n\t v_readlane_b32 s29, %[vp], 2\n\t"
            "v_readlane_b32 s30, %[vl], 3\n\t v_readlane_b32 s31, %[vp], 3\n\t"
            SYM("s24", "s25", "s[32:33]")
            SYM("s26", "s27", "s[34:35]")
            SYM("s28", "s29", "s[36:37]")
            SYM("s30", "s31", "s[38:39]")
            "s_and_b64 s[32:33], s[32:33], s[34:35]\n\t"
            "s_and_b64 s[36:37], s[36:37], s[38:39]\n\t"
            "s_and_b64 s[32:33], s[32:33], s[36:37]\n\t"
            "s_mov_b32 %[ok], s32\n\t"
            "v_mov_b32 %[d0], v4\n\t v_mov_b32 %[d1], v5\n\t v_mov_b32 %[r0], v6\n\t v_or_b32 %[r1], 0xf0000000, v7\n\t"
            : [d0] "+v"(d0), [d1] "+v"(d1), [r0] "+v"(r0), [r1] "+v"(r1), [ok] "=s"(ok)
            : [vl] "v"(vl), [vp] "v"(vp)
            : "vcc", "v4", "v5", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "s20", "s21", "s24", "s25", "s26", "s27", "s28", "s29", "s30",
              "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39");
        okacc &= ok;
        if (okacc == 0x12345u) break;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = d0 + d1 + r0 + r1 + okacc; }
}

int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 64));
    uint64_t h[2];
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(spec_kernel, dim3(1), dim3(64), 0, 0, d, 1000u, 5000u, 4096); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("speculative block of 4, state back in SGPRs after every block: %.1f ticks / symbol (%.1f / block)\n", h[0] / 4096.0 / 4, h[0] / 4096.0);
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(spec_kernel_vstate, dim3(1), dim3(64), 0, 0, d, 1000u, 5000u, 4096); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("speculative block of 4, state stays in VGPRs:                  %.1f ticks / symbol (%.1f / block)\n", h[0] / 4096.0 / 4, h[0] / 4096.0);
    return 0;
}
