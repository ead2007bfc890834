// Throughput probes (per CU, 4/8 waves): v_mad_i64_i32 vs v_fma_f64 vs v_mad_u32_u24 vs ds_read_b64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define PROBE(name, body)                                                               \
    __global__ __launch_bounds__(512) void name(uint64_t* out, uint32_t seed) {         \
        uint32_t v = threadIdx.x + seed, w = v * 3 + 1;                                  \
        __shared__ double lds[1024];                                                     \
        lds[threadIdx.x] = v; lds[threadIdx.x + 512] = w; __syncthreads();               \
        uint32_t addr = (threadIdx.x & 63) * 8;                                          \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                      \
        for (int it = 0; it < 16; ++it)                                                  \
            asm volatile(REP64(body) : "+v"(v), "+v"(w) : "v"(addr) : "vcc", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25"); \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                      \
        if ((threadIdx.x & 63) == 0) { out[threadIdx.x / 64] = t1 - t0; out[16] = v + w; } \
    }
// 4 independent accumulators per body
PROBE(p_mad64, "v_mad_i64_i32 v[10:11], vcc, %0, %1, v[10:11]\n v_mad_i64_i32 v[12:13], vcc, %0, %1, v[12:13]\n v_mad_i64_i32 v[14:15], vcc, %0, %1, v[14:15]\n v_mad_i64_i32 v[16:17], vcc, %0, %1, v[16:17]\n")
PROBE(p_fma64, "v_fma_f64 v[10:11], v[18:19], v[20:21], v[10:11]\n v_fma_f64 v[12:13], v[18:19], v[20:21], v[12:13]\n v_fma_f64 v[14:15], v[18:19], v[20:21], v[14:15]\n v_fma_f64 v[16:17], v[18:19], v[20:21], v[16:17]\n")
PROBE(p_mad24, "v_mad_i32_i24 v10, %0, %1, v10\n v_mad_i32_i24 v12, %0, %1, v12\n v_mad_i32_i24 v14, %0, %1, v14\n v_mad_i32_i24 v16, %0, %1, v16\n")
PROBE(p_fma32, "v_fma_f32 v10, %0, %1, v10\n v_fma_f32 v12, %0, %1, v12\n v_fma_f32 v14, %0, %1, v14\n v_fma_f32 v16, %0, %1, v16\n")
PROBE(p_dsr64, "ds_read_b64 v[10:11], %2\n ds_read_b64 v[12:13], %2 offset:512\n ds_read_b64 v[14:15], %2 offset:1024\n ds_read_b64 v[16:17], %2 offset:1536\n s_waitcnt lgkmcnt(0)\n")
PROBE(p_fma64_lds, "ds_read_b64 v[22:23], %2\n ds_read_b64 v[24:25], %2 offset:512\n s_waitcnt lgkmcnt(0)\n v_fma_f64 v[10:11], v[22:23], v[20:21], v[10:11]\n v_fma_f64 v[12:13], v[24:25], v[20:21], v[12:13]\n")
typedef void (*kern_t)(uint64_t*, uint32_t);
int main() {
    uint64_t* d; hipMalloc(&d, 256);
    struct { const char* n; kern_t k; int per; } P[] = {{"v_mad_i64_i32", p_mad64, 4}, {"v_fma_f64", p_fma64, 4}, {"v_mad_i32_i24", p_mad24, 4},
        {"v_fma_f32", p_fma32, 4}, {"ds_read_b64 (+wait per 4)", p_dsr64, 4}, {"2x(ds_read_b64 + v_fma_f64)", p_fma64_lds, 2}};
    for (int threads : {64, 256, 512})
        for (auto& p : P) {
            uint64_t h[8];
            for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(p.k, dim3(1), dim3(threads), 0, 0, d, 5u); hipDeviceSynchronize(); }
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            double n_instr = 16.0 * 64 * p.per;
            printf("%4d threads  %-30s %7.2f ticks per wave-instr (wave0), i.e. %6.2f per instr per CU\n", threads, p.n, h[0] / n_instr, h[0] / n_instr / (threads / 64));
        }
    return 0;
}
