// Whole-chip issue throughput of small f32 MFMAs beside VALU work (MI355X): what one SIMD sustains with
// 1, 2 or 4 resident waves.  Timed with HIP events over a long loop; cycles assume 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_thr.hip -o tools/ubench/mfma_thr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: 4x4x1 x NM ; MODE 1: 16x16x4 x NM ; NV v_fma per body
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(256) void body_kernel(float* out, float seed, int iters) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = seed + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < (NM > NV ? NM : NV); ++j) {
            if (j < NM) {
                if (MODE == 0) c[j & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j & 7], 0, 0, 0);
                else c[j & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j & 7], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < (NV + (NM > 0 ? NM : 1) - 1) / (NM > 0 ? NM : 1); ++q) {
                const int idx = (j * 3 + q) & 15;
                if (NM == 0 || j * ((NV + NM - 1) / NM) + q < NV) { v[idx] = __fmaf_rn(v[idx], a, b); }
            }
        }
    }
    f32x4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    float vs = 0;
    for (int j = 0; j < 16; ++j) vs += v[j];
    if (s[0] + s[1] + s[2] + s[3] + vs == 12345.f) out[0] = 1;
}

template <int MODE, int NM, int NV>
static int run(const char* name, float* d, int waves_per_simd) {
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((body_kernel<MODE, NM, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.5f, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((body_kernel<MODE, NM, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.5f, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc_per_body_per_simd = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
    const double fma_per_body = NM * (MODE == 0 ? 256.0 : 1024.0) + NV * 64.0;
    const double tflops = 2.0 * fma_per_body * iters * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-44s waves/SIMD %d: %8.3f ms  %7.1f cyc/body/SIMD  %6.1f TFLOP/s\n", name, waves_per_simd, ms, cyc_per_body_per_simd, tflops);
    return 0;
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, 64));
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 8, 0>("8 x mfma 4x4x1", d, w);
        run<1, 8, 0>("8 x mfma 16x16x4", d, w);
        run<0, 0, 16>("16 x v_fma", d, w);
        run<0, 8, 8>("8 x mfma 4x4x1 + 8 v_fma", d, w);
        run<0, 8, 16>("8 x mfma 4x4x1 + 16 v_fma", d, w);
        run<0, 8, 32>("8 x mfma 4x4x1 + 32 v_fma", d, w);
        run<1, 8, 32>("8 x mfma 16x16x4 + 32 v_fma", d, w);
        run<1, 8, 64>("8 x mfma 16x16x4 + 64 v_fma", d, w);
    }
    return 0;
}
