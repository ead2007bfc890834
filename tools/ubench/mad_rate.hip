// Issue cost of the multiply-add flavours the ARM's integer layers could use (MI355X, one wave, independent instructions).
//   hipcc --offload-arch=gfx950 -O2 -o mad_rate tools/ubench/mad_rate.hip && ./mad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

#define PROBE(name, body)                                                              \
    __global__ __launch_bounds__(64) void name(uint64_t* out, uint32_t seed) {        \
        uint32_t v = threadIdx.x + seed, w = v * 3 + 1;                                \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                    \
        asm volatile(REP256(body) : "+v"(v), "+v"(w) :: "vcc", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25"); \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                    \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v + w; }                    \
    }

// four independent instructions per group
PROBE(p_mad_i64_i32, "v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_i64_i32 v[12:13], vcc, %0, %1, v[12:13]\n v_mad_i64_i32 v[14:15], vcc, %0, %1, v[14:15]\n v_mad_i64_i32 v[16:17], vcc, %0, %1, v[16:17]\n")
PROBE(p_mad_u64_u32, "v_mad_u64_u32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_u64_u32 v[12:13], vcc, %0, %1, v[12:13]\n v_mad_u64_u32 v[14:15], vcc, %0, %1, v[14:15]\n v_mad_u64_u32 v[16:17], vcc, %0, %1, v[16:17]\n")
PROBE(p_mad_i32_i24, "v_mad_i32_i24 v10, %0, %1, v10\n v_mad_i32_i24 v12, %0, %1, v12\n v_mad_i32_i24 v14, %0, %1, v14\n v_mad_i32_i24 v16, %0, %1, v16\n")
PROBE(p_mul_lo_u32, "v_mul_lo_u32 v10, %0, %1\n v_mul_lo_u32 v12, %0, %1\n v_mul_lo_u32 v14, %0, %1\n v_mul_lo_u32 v16, %0, %1\n")
PROBE(p_mul_hi_i32, "v_mul_hi_i32 v10, %0, %1\n v_mul_hi_i32 v12, %0, %1\n v_mul_hi_i32 v14, %0, %1\n v_mul_hi_i32 v16, %0, %1\n")
PROBE(p_mul_hi_i24, "v_mul_hi_i32_i24 v10, %0, %1\n v_mul_hi_i32_i24 v12, %0, %1\n v_mul_hi_i32_i24 v14, %0, %1\n v_mul_hi_i32_i24 v16, %0, %1\n")
PROBE(p_dot2_i16, "v_dot2_i32_i16 v10, %0, %1, v10\n v_dot2_i32_i16 v12, %0, %1, v12\n v_dot2_i32_i16 v14, %0, %1, v14\n v_dot2_i32_i16 v16, %0, %1, v16\n")
PROBE(p_dot4_i8, "v_dot4_i32_i8 v10, %0, %1, v10\n v_dot4_i32_i8 v12, %0, %1, v12\n v_dot4_i32_i8 v14, %0, %1, v14\n v_dot4_i32_i8 v16, %0, %1, v16\n")
PROBE(p_fma_f64, "v_fma_f64 v[10:11], v[18:19], v[20:21], v[10:11]\n v_fma_f64 v[12:13], v[18:19], v[20:21], v[12:13]\n v_fma_f64 v[14:15], v[18:19], v[20:21], v[14:15]\n v_fma_f64 v[16:17], v[18:19], v[20:21], v[16:17]\n")
PROBE(p_fma_f32, "v_fma_f32 v10, %0, %1, v10\n v_fma_f32 v12, %0, %1, v12\n v_fma_f32 v14, %0, %1, v14\n v_fma_f32 v16, %0, %1, v16\n")
PROBE(p_add_u32, "v_add_u32 v10, %0, v10\n v_add_u32 v12, %0, v12\n v_add_u32 v14, %0, v14\n v_add_u32 v16, %0, v16\n")
PROBE(p_lshl_add_u64, "v_lshl_add_u64 v[10:11], v[18:19], 0, v[10:11]\n v_lshl_add_u64 v[12:13], v[18:19], 0, v[12:13]\n v_lshl_add_u64 v[14:15], v[18:19], 0, v[14:15]\n v_lshl_add_u64 v[16:17], v[18:19], 0, v[16:17]\n")
// one dependent chain
PROBE(p_mad_i64_dep, "v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n")
PROBE(p_mad_i24_dep, "v_mad_i32_i24 v10, %0, %1, v10\n v_mad_i32_i24 v10, %0, %1, v10\n v_mad_i32_i24 v10, %0, %1, v10\n v_mad_i32_i24 v10, %0, %1, v10\n")

typedef void (*kern_t)(uint64_t*, uint32_t);
int main() {
    uint64_t* d; CHECK(hipMalloc(&d, 64));
    struct { const char* n; kern_t k; } P[] = {
        {"v_mad_i64_i32 (independent)", p_mad_i64_i32}, {"v_mad_u64_u32", p_mad_u64_u32}, {"v_mad_i32_i24", p_mad_i32_i24}, {"v_mul_lo_u32", p_mul_lo_u32},
        {"v_mul_hi_i32", p_mul_hi_i32}, {"v_mul_hi_i32_i24", p_mul_hi_i24}, {"v_dot2_i32_i16", p_dot2_i16}, {"v_dot4_i32_i8", p_dot4_i8}, {"v_fma_f64", p_fma_f64},
        {"v_fma_f32", p_fma_f32}, {"v_add_u32", p_add_u32}, {"v_lshl_add_u64", p_lshl_add_u64}, {"v_mad_i64_i32 (dependent chain)", p_mad_i64_dep}, {"v_mad_i32_i24 (dependent chain)", p_mad_i24_dep}};
    for (auto& p : P) {
        uint64_t h[2];
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, d, 5u); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%-36s %7.2f s_memtime ticks / instruction\n", p.n, h[0] / 1024.0);
    }
    return 0;
}
