// r06: are the FATTER f32 MFMA shapes exact fmaf chains?  (r05 verdict, float path item 3a.)
//   v_mfma_f32_16x16x4_f32 (K = 4 per issue), v_mfma_f32_32x32x2_f32 (K = 2): is the internal K accumulation
//       c' = fma(a[K-1], b[K-1], ... fma(a1, b1, fma(a0, b0, c)))   (ascending k, one rounding per product)?
//   v_mfma_f32_16x16x1_4b_f32, v_mfma_f32_32x32x1_2b_f32 (K = 1, 16 / 32 outputs per lane): one fma per output like 4x4x1?
// For every output word the probe evaluates the hypotheses on the host (std::fmaf) and counts bitwise matches:
//   asc   sequential fma chain, k ascending          desc  sequential fma chain, k descending
//   exact exact sum of c and all products (long double / two-sum) rounded ONCE to f32 ("fused dot")
//   pair  (a0 b0 + a1 b1) + (a2 b2 + a3 b3) products rounded, tree sum, then + c
// Layouts are checked first with small integers (exact in every order).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_probe_k.hip -o tools/ubench/mfma_probe_k
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// n_steps chained issues: operands a[s][lane], b[s][lane]; accumulator c0[lane][R]
__global__ void k_16x16x4(const float* a, const float* b, const float* c0, int n_steps, float* d) {
    const int l = threadIdx.x;
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = c0[l * 4 + r];
    for (int s = 0; s < n_steps; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s * 64 + l], b[s * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void k_32x32x2(const float* a, const float* b, const float* c0, int n_steps, float* d) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = c0[l * 16 + r];
    for (int s = 0; s < n_steps; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s * 64 + l], b[s * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}
__global__ void k_16x16x1(const float* a, const float* b, const float* c0, int n_steps, float* d) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = c0[l * 16 + r];
    for (int s = 0; s < n_steps; ++s) c = __builtin_amdgcn_mfma_f32_16x16x1f32(a[s * 64 + l], b[s * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}
__global__ void k_32x32x1(const float* a, const float* b, const float* c0, int n_steps, float* d) {
    const int l = threadIdx.x;
    f32x32 c;
    for (int r = 0; r < 32; ++r) c[r] = c0[l * 32 + r];
    for (int s = 0; s < n_steps; ++s) c = __builtin_amdgcn_mfma_f32_32x32x1f32(a[s * 64 + l], b[s * 64 + l], c, 0, 0, 0);
    for (int r = 0; r < 32; ++r) d[l * 32 + r] = c[r];
}

// ---- issue rates: 4 independent accumulators ----
template <int SHAPE>
__global__ void rate_kernel(uint64_t* out, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    float s = 0;
    if constexpr (SHAPE == 0) {
        f32x4 c[4] = {};
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
        s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (SHAPE == 1) {
        f32x16 c[4] = {};
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
        s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (SHAPE == 2) {
        f32x16 c[4] = {};
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c[j], 0, 0, 0);
        s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else if constexpr (SHAPE == 3) {
        f32x32 c[2] = {};
        for (int i = 0; i < 128; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, c[j], 0, 0, 0);
        s = c[0][0] + c[1][1];
    } else {
        f32x4 c[4] = {};
        for (int i = 0; i < 64; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
        s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (s == 12345.f) out[1] = 1;
}

struct Shape {
    const char* name;
    int M, N, K, B, regs;  // B blocks
    // lane of A[blk][i][k], B[blk][k][j]; (lane, reg) of D[blk][i][j]
    int a_lane(int blk, int i, int k) const { return M == 32 ? (B == 2 ? 32 * blk + i : 32 * k + i) : (B == 4 ? 16 * blk + i : 16 * k + i); }
    int b_lane(int blk, int k, int j) const { return a_lane(blk, j, k); }
    void d_at(int blk, int i, int j, int* lane, int* reg) const {
        if (M == 16) { *lane = 16 * (i / 4) + j; *reg = 4 * blk + i % 4; }
        else { *lane = 32 * ((i / 4) % 2) + j; *reg = 16 * blk + 4 * (i / 8) + i % 4; }
    }
};

static float frand() { return (float)((double)rand() / RAND_MAX * 4.0 - 2.0); }
static float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// c + sum of products, rounded once (products of two floats are exact in double; the sum of <= 5 doubles in long double with a
// two-sum compensation is exact enough to decide the f32 rounding except in pathological ties, which the counts tolerate)
static float fused_dot(float c, const float* a, const float* b, int K) {
    long double s = c;
    for (int k = 0; k < K; ++k) s += (long double)((double)a[k] * (double)b[k]);
    return (float)s;
}

int main() {
    const Shape shapes[4] = {{"16x16x4 (1 block)", 16, 16, 4, 1, 4}, {"32x32x2 (1 block)", 32, 32, 2, 1, 16},
                             {"16x16x1 (4 blocks)", 16, 16, 1, 4, 16}, {"32x32x1 (2 blocks)", 32, 32, 1, 2, 32}};
    const int S = 12;  // chained issues
    float *da, *db, *dc, *dd;
    CHECK(hipMalloc(&da, S * 256)); CHECK(hipMalloc(&db, S * 256)); CHECK(hipMalloc(&dc, 64 * 32 * 4)); CHECK(hipMalloc(&dd, 64 * 32 * 4));
    for (int si = 0; si < 4; ++si) {
        const Shape& sh = shapes[si];
        std::vector<float> a(S * 64), b(S * 64), c0(64 * sh.regs), d(64 * sh.regs);
        auto run = [&](int n_steps) -> int {
            CHECK(hipMemcpy(da, a.data(), S * 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b.data(), S * 256, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dc, c0.data(), c0.size() * 4, hipMemcpyHostToDevice));
            if (si == 0) hipLaunchKernelGGL(k_16x16x4, dim3(1), dim3(64), 0, 0, da, db, dc, n_steps, dd);
            if (si == 1) hipLaunchKernelGGL(k_32x32x2, dim3(1), dim3(64), 0, 0, da, db, dc, n_steps, dd);
            if (si == 2) hipLaunchKernelGGL(k_16x16x1, dim3(1), dim3(64), 0, 0, da, db, dc, n_steps, dd);
            if (si == 3) hipLaunchKernelGGL(k_32x32x1, dim3(1), dim3(64), 0, 0, da, db, dc, n_steps, dd);
            CHECK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
            return 0;
        };
        // ---- layout with small integers
        for (int l = 0; l < S * 64; ++l) { a[l] = (float)(1 + (l * 7) % 13); b[l] = (float)(1 + (l * 5) % 11); }
        for (auto& f : c0) f = 0.f;
        if (run(1)) return 1;
        int bad = 0;
        for (int blk = 0; blk < sh.B; ++blk) for (int i = 0; i < sh.M; ++i) for (int j = 0; j < sh.N; ++j) {
            float want = 0;
            for (int k = 0; k < sh.K; ++k) want += a[sh.a_lane(blk, i, k)] * b[sh.b_lane(blk, k, j)];
            int ln, rg; sh.d_at(blk, i, j, &ln, &rg);
            if (d[ln * sh.regs + rg] != want) ++bad;
        }
        printf("%s layout: %s (%d mismatches)\n", sh.name, bad ? "NOT as assumed" : "as assumed", bad);
        if (bad) continue;
        // ---- exactness by class
        const char* cls_name[4] = {"normal range", "wide exponents (2^-20 .. 2^20), cancellation", "arbitrary bit patterns", "subnormal / underflowing"};
        for (int cls = 0; cls < 4; ++cls) {
            long total = 0, m_asc = 0, m_desc = 0, m_exact = 0, m_pair = 0, m_any = 0, nan_skipped = 0;
            for (int trial = 0; trial < 60; ++trial) {
                srand(9000 + 17 * cls + trial);
                for (auto& f : a) f = frand();
                for (auto& f : b) f = frand();
                for (auto& f : c0) f = frand();
                if (cls == 1) {
                    for (auto& f : a) f = ldexpf(f, rand() % 41 - 20);
                    for (auto& f : b) f = ldexpf(f, rand() % 41 - 20);
                    for (auto& f : c0) f = (rand() % 3 == 0) ? 0.f : ldexpf(f, rand() % 41 - 20);
                } else if (cls == 2) {
                    auto ur = []() { return (uint32_t)rand() * 2654435761u ^ ((uint32_t)rand() << 11); };
                    for (auto& f : a) f = bitsf((ur() & 0x80ffffffu) | ((uint32_t)(90 + rand() % 76) << 23));
                    for (auto& f : b) f = bitsf((ur() & 0x80ffffffu) | ((uint32_t)(90 + rand() % 76) << 23));
                    for (auto& f : c0) f = bitsf((ur() & 0x80ffffffu) | ((uint32_t)(90 + rand() % 76) << 23));
                } else if (cls == 3) {
                    for (auto& f : a) f *= (rand() & 1) ? 1e-22f : 1e-19f;
                    for (auto& f : b) f *= (rand() & 1) ? 1e-22f : 1e-19f;
                    for (auto& f : c0) { int r = rand() % 3; f = r == 0 ? 0.0f : f * (r == 1 ? 1e-41f : 1e-37f); }
                }
                if (run(S)) return 1;
                for (int blk = 0; blk < sh.B; ++blk) for (int i = 0; i < sh.M; ++i) for (int j = 0; j < sh.N; ++j) {
                    int ln, rg; sh.d_at(blk, i, j, &ln, &rg);
                    const float got = d[ln * sh.regs + rg];
                    float h_asc = c0[ln * sh.regs + rg], h_desc = h_asc, h_exact = h_asc, h_pair = h_asc;
                    for (int s = 0; s < S; ++s) {
                        float av[4], bv[4];
                        for (int k = 0; k < sh.K; ++k) { av[k] = a[s * 64 + sh.a_lane(blk, i, k)]; bv[k] = b[s * 64 + sh.b_lane(blk, k, j)]; }
                        for (int k = 0; k < sh.K; ++k) h_asc = fmaf(av[k], bv[k], h_asc);
                        for (int k = sh.K - 1; k >= 0; --k) h_desc = fmaf(av[k], bv[k], h_desc);
                        h_exact = fused_dot(h_exact, av, bv, sh.K);
                        if (sh.K == 4) { volatile float p0 = av[0] * bv[0], p1 = av[1] * bv[1], p2 = av[2] * bv[2], p3 = av[3] * bv[3]; volatile float q0 = p0 + p1, q1 = p2 + p3; volatile float q = q0 + q1; h_pair = h_pair + q; }
                        else if (sh.K == 2) { volatile float p0 = av[0] * bv[0], p1 = av[1] * bv[1]; volatile float q = p0 + p1; h_pair = h_pair + q; }
                        else h_pair = fmaf(av[0], bv[0], h_pair);
                    }
                    if (got != got) { ++nan_skipped; continue; }
                    ++total;
                    const uint32_t g = fbits(got);
                    const bool e0 = g == fbits(h_asc), e1 = g == fbits(h_desc), e2 = g == fbits(h_exact), e3 = g == fbits(h_pair);
                    m_asc += e0; m_desc += e1; m_exact += e2; m_pair += e3; m_any += (e0 || e1 || e2 || e3);
                }
            }
            printf("  %-46s of %8ld words: == asc fma chain %8ld | desc %8ld | fused-dot (one rounding) %8ld | product-rounded tree %8ld | none of them %8ld  (NaN skipped %ld)\n",
                   cls_name[cls], total, m_asc, m_desc, m_exact, m_pair, total - m_any, nan_skipped);
        }
    }
    // ---- issue intervals (one wave, 4 independent accumulators)
    uint64_t* dout; CHECK(hipMalloc(&dout, 64)); CHECK(hipMemset(dout, 0, 64));
    uint64_t ho[2];
    const char* rn[5] = {"16x16x4", "32x32x2", "16x16x1 4b", "32x32x1 2b", "4x4x1 16b"};
    for (int rep = 0; rep < 2; ++rep)
        for (int r = 0; r < 5; ++r) {
            if (r == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(1), dim3(64), 0, 0, dout, 1.5f);
            if (r == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(1), dim3(64), 0, 0, dout, 1.5f);
            if (r == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(1), dim3(64), 0, 0, dout, 1.5f);
            if (r == 3) hipLaunchKernelGGL((rate_kernel<3>), dim3(1), dim3(64), 0, 0, dout, 1.5f);
            if (r == 4) hipLaunchKernelGGL((rate_kernel<4>), dim3(1), dim3(64), 0, 0, dout, 1.5f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost));
            if (rep) printf("rate %-12s %7.2f cycles per issue (one wave, independent accumulators)\n", rn[r], (double)ho[0] / 256.0);
        }
    return 0;
}
