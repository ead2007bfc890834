// Does the decoder wave lose time to the producer wave that shares its SIMD?  One 512-thread workgroup: wave 0 runs a dependent
// VALU -> SALU -> VALU chain (the shape of the range-decoder recurrence) at priority 3; the waves selected by `busy_mask` run
// independent v_mad_u64_u32 / f64 fma streams (the producers' instruction mix) until wave 0 is done.  Prints the SIMD of every
// wave (HW_ID) and ticks per chain iteration for a few masks.
//     hipcc --offload-arch=gfx950 -O3 -o simd_share simd_share.hip && ./simd_share
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(512) void k(uint32_t busy_mask, int iters, uint32_t* out, int kind) {
    __shared__ uint32_t done;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (lane == 0) out[8 + wave] = hw;
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        uint32_t s = 12345u;
        uint64_t v = lane + 1;
        uint32_t vl = lane;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            // 2 VALU (one quarter-rate), a hand-over, 2 SALU: roughly a third of a symbol's chain
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                "v_xor_b32 %2, %2, %3\n\t"
                "s_nop 0\n\t"
                "v_readfirstlane_b32 %1, %2\n\t"
                "s_add_u32 %1, %1, 7\n\t"
                "s_xor_b32 %1, %1, 0x55\n\t"
                : "+v"(v), "+s"(s), "+v"(vl) : "v"(static_cast<uint32_t>(v >> 32)) : "vcc", "scc");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) { out[0] = static_cast<uint32_t>(t1 - t0); out[1] = s + static_cast<uint32_t>(v); }
        __atomic_store_n(&done, 1u, __ATOMIC_RELEASE);
    } else if ((busy_mask >> wave) & 1u) {
        uint64_t a = lane, b = lane * 3, c = lane * 5, d = lane * 7;
        double f0 = lane, f1 = lane + 0.5;
        uint32_t x = lane | 1u;
        while (__atomic_load_n(&done, __ATOMIC_ACQUIRE) == 0) {
            for (int r = 0; r < 16; ++r) {
                if (kind == 0) {
                    a += static_cast<uint64_t>(x) * static_cast<uint32_t>(b);
                    b += static_cast<uint64_t>(x) * static_cast<uint32_t>(c);
                    c += static_cast<uint64_t>(x) * static_cast<uint32_t>(d);
                    d += static_cast<uint64_t>(x) * static_cast<uint32_t>(a);
                } else {
                    f0 = fma(f0, 1.0000001, f1);
                    f1 = fma(f1, 0.9999999, f0);
                }
            }
        }
        if (lane == 0) out[16 + wave] = static_cast<uint32_t>(a + b + c + d) + static_cast<uint32_t>(f0 + f1);
    }
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 64 * 4);
    uint32_t h[64];
    const int iters = 200000;
    const uint32_t masks[] = {0x00, 0xfe, 0xee, 0x10, 0x0e, 0x02, 0x20, 0xf0};
    for (int kind = 0; kind < 2; ++kind)
        for (uint32_t m : masks) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, m, iters, d, kind);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s busy waves mask 0x%02x: %.1f ticks / iteration\n", kind ? "f64 fma" : "mad_u64", m, static_cast<double>(h[0]) / iters);
        }
    printf("SIMD of waves 0..7:");
    for (int w = 0; w < 8; ++w) printf(" %u", (h[8 + w] >> 4) & 3u);
    printf("   (HW_ID wave slot:");
    for (int w = 0; w < 8; ++w) printf(" %u", h[8 + w] & 15u);
    printf(")\n");
    return 0;
}
