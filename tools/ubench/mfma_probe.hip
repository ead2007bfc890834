// Probes for the f32 / i8 multi-block MFMAs the round-2 kernels lean on (MI355X, gfx950):
//   1. lane layout and CBSZ / ABID broadcast of v_mfma_f32_4x4x1_16b_f32
//   2. bitwise equality of its accumulate with fmaf (one rounding per product, chain in issue order)
//   3. issue interval (independent accumulators), dependent latency, and co-issue with VALU fma
//   4. the same rates for v_mfma_i32_4x4x4_16b_i8 and v_mfma_i32_16x16x64_i8
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void layout_kernel(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

// chain of K accumulates with broadcast weights: D[r] = fma(w[k][r], x[k][lane], D[r]) for k ascending
__global__ void exact_kernel(const float* w /*[K][4]*/, const float* x /*[K][64]*/, const float* c0 /*[64][4]*/, int K, float* d_mfma, float* d_valu) {
    const int l = threadIdx.x;
    f32x4 c;
    float v[4];
    for (int r = 0; r < 4; ++r) { c[r] = c0[l * 4 + r]; v[r] = c0[l * 4 + r]; }
    for (int k = 0; k < K; ++k) {
        const float av = w[k * 4 + (l & 3)];  // lane 4b+i holds A[i]
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x[k * 64 + l], c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) v[r] = __fmaf_rn(w[k * 4 + r], x[k * 64 + l], v[r]);
    }
    for (int r = 0; r < 4; ++r) { d_mfma[l * 4 + r] = c[r]; d_valu[l * 4 + r] = v[r]; }
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// ---- rates (per wave, s_memtime ticks = shader cycles) ----
__global__ void rate_f32_indep(uint64_t* out, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[blockIdx.x * 2 + 1] = 1;
}
template <int NACC>
__global__ void rate_f32_chain(uint64_t* out, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x4 c[NACC];
    for (int j = 0; j < NACC; ++j) c[j] = f32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 512 / NACC; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c[0];
    for (int j = 1; j < NACC; ++j) s += c[j];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[blockIdx.x * 2 + 1] = 1;
}
// NV VALU fmas between consecutive MFMAs (8 accumulators): what does the VALU work cost beside the matrix pipe?
template <int NV>
__global__ void rate_f32_mixed(uint64_t* out, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f32x4{0, 0, 0, 0};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) { v[(j + q) & 7] = __fmaf_rn(v[(j + q) & 7], a, b); asm volatile("" : "+v"(v[(j + q) & 7])); }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    float vs = 0;
    for (int j = 0; j < 8; ++j) vs += v[j];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] + vs == 12345.f) out[blockIdx.x * 2 + 1] = 1;
}
__global__ void rate_valu_only(uint64_t* out, float seed) {
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = __fmaf_rn(v[j], a, b); asm volatile("" : "+v"(v[j])); }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float vs = 0;
    for (int j = 0; j < 8; ++j) vs += v[j];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (vs == 12345.f) out[blockIdx.x * 2 + 1] = 1;
}
__global__ void rate_i8_4x4x4(uint64_t* out, int seed) {
    int a = seed + threadIdx.x, b = seed * 3;
    i32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = i32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, c[j], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    i32x4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345) out[blockIdx.x * 2 + 1] = 1;
}
__global__ void rate_i8_16x16x64(uint64_t* out, int seed) {
    i32x4 a = {seed + (int)threadIdx.x, seed, seed * 3, seed * 5}, b = {seed * 7, seed, seed, seed};
    i32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = i32x4{0, 0, 0, 0};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[j], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    i32x4 s = c[0];
    for (int j = 1; j < 8; ++j) s += c[j];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345) out[blockIdx.x * 2 + 1] = 1;
}
// i8 layout: 4x4x4 16 blocks: D[i][j] = sum_k A[i][k] B[k][j]; lane 4b+i holds A[i][0..3] as bytes, lane 4b+j holds B[0..3][j]
__global__ void layout_i8_kernel(const int* a, const int* b, int* d) {
    const int l = threadIdx.x;
    i32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

static float frand() { return (float)((double)rand() / RAND_MAX * 4.0 - 2.0); }

int main() {
    float *da, *db, *dd;
    CHECK(hipMalloc(&da, 256)); CHECK(hipMalloc(&db, 256)); CHECK(hipMalloc(&dd, 1024));
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (float)(100 * (l + 1)); }
    CHECK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
    // ---- 1. layout ----
    {
        hipLaunchKernelGGL((layout_kernel<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
        CHECK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[(l & ~3) + r] * hb[l]) ++bad;
        printf("layout cbsz=0: D[lane][r] == A[4*(lane/4)+r] * B[lane]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
        if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
        hipLaunchKernelGGL((layout_kernel<4, 5>), dim3(1), dim3(64), 0, 0, da, db, dd);
        CHECK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[4 * 5 + r] * hb[l]) ++bad;
        printf("layout cbsz=4 abid=5: D[lane][r] == A[4*5+r] * B[lane]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
        if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
        hipLaunchKernelGGL((layout_kernel<2, 3>), dim3(1), dim3(64), 0, 0, da, db, dd);
        CHECK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
        bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hd[l * 4 + r] != ha[4 * (((l / 4) & ~3) + 3) + r] * hb[l]) ++bad;
        printf("layout cbsz=2 abid=3: D[lane][r] == A[4*(4*(block/4)+3)+r] * B[lane]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
        if (bad) for (int l = 0; l < 64; l += 7) printf("  lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    }
    // ---- 2. exactness ----
    {
        const int K = 57;
        std::vector<float> w(K * 4), x(K * 64), c0(256), m(256), v(256);
        float *dw, *dx, *dc, *dm, *dv;
        CHECK(hipMalloc(&dw, K * 16)); CHECK(hipMalloc(&dx, K * 256)); CHECK(hipMalloc(&dc, 1024)); CHECK(hipMalloc(&dm, 1024)); CHECK(hipMalloc(&dv, 1024));
        long bad = 0, total = 0;
        for (int trial = 0; trial < 200; ++trial) {
            srand(1234 + trial);
            const float scale = trial % 3 == 0 ? 1e-3f : (trial % 3 == 1 ? 1.0f : 300.0f);
            for (auto& f : w) f = frand() * scale;
            for (auto& f : x) f = frand();
            for (auto& f : c0) f = trial % 5 == 0 ? 0.0f : frand();
            CHECK(hipMemcpy(dw, w.data(), K * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, x.data(), K * 256, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dc, c0.data(), 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, dw, dx, dc, K, dm, dv);
            CHECK(hipMemcpy(m.data(), dm, 1024, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(v.data(), dv, 1024, hipMemcpyDeviceToHost));
            for (int i = 0; i < 256; ++i) { ++total; if (memcmp(&m[i], &v[i], 4)) ++bad; }
        }
        printf("exactness: 4x4x1 MFMA chain (K=57) vs __fmaf_rn chain: %ld / %ld words differ\n", bad, total);
    }
    // ---- 2b. exactness by operand class (r05): the float stages' claim "a chain of v_mfma_f32_4x4x1 IS the oracle's fmaf chain"
    // for what a bit-flipped network payload can produce: subnormal operands, products that underflow, accumulators of +-0 with zero
    // products of both signs, +-inf, NaN, values near FLT_MAX.  Three chains per case: the MFMA, __fmaf_rn on the vector ALU
    // (the unfused kernels), std::fmaf on the host (what oracle/cc_oracle.c calls).  "differ" = bitwise; "non-NaN" leaves out the
    // words that are NaN on both sides (IEEE 754 does not fix a NaN's payload or sign).
    {
        const int K = 57;
        std::vector<float> w(K * 4), x(K * 64), c0(256), m(256), v(256), h(256);
        float *dw, *dx, *dc, *dm, *dv;
        CHECK(hipMalloc(&dw, K * 16)); CHECK(hipMalloc(&dx, K * 256)); CHECK(hipMalloc(&dc, 1024)); CHECK(hipMalloc(&dm, 1024)); CHECK(hipMalloc(&dv, 1024));
        auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        auto urand = []() { return (uint32_t)rand() * 2654435761u ^ ((uint32_t)rand() << 11); };
        auto sgn = [&]() { return (rand() & 1) ? -1.0f : 1.0f; };
        struct Cls { const char* name; int id; };
        const Cls classes[] = {{"subnormal inputs (w or x below 2^-126)", 0}, {"products that underflow, small accumulators", 1},
                               {"accumulators +-0, zero products of both signs", 2}, {"+-inf among the operands", 3},
                               {"NaN among the operands", 4}, {"near FLT_MAX: overflowing products and sums", 5},
                               {"arbitrary bit patterns (any exponent, any class)", 6}, {"cancellation: sums that return to 0 or to subnormals", 7}};
        for (const Cls& cl : classes) {
            long bad_mv = 0, bad_mv_nn = 0, bad_mh = 0, bad_mh_nn = 0, total = 0, n_nan = 0, n_sub = 0, n_inf = 0, n_negzero = 0;
            for (int trial = 0; trial < 100; ++trial) {
                srand(7000 + 131 * cl.id + trial);
                for (auto& f : w) f = frand();
                for (auto& f : x) f = frand();
                for (auto& f : c0) f = frand();
                switch (cl.id) {
                    case 0:
                        for (auto& f : w) if (rand() % 3 == 0) f = bits((urand() & 0x807fffffu));             // subnormal, either sign
                        for (auto& f : x) if (rand() % 3 == 0) f = bits((urand() & 0x807fffffu));
                        for (auto& f : c0) { int r = rand() % 4; if (r == 0) f = bits(urand() & 0x807fffffu); else if (r == 1) f = 0.0f; else if (r == 2) f *= 1e-38f; }
                        if (trial & 1) { for (auto& f : w) f *= 1e-30f; }                                       // normal x subnormal -> far below
                        break;
                    case 1:
                        for (auto& f : w) f *= (rand() & 1) ? 1e-22f : 1e-19f;
                        for (auto& f : x) f *= (rand() & 1) ? 1e-22f : 1e-19f;
                        for (auto& f : c0) { int r = rand() % 3; f = r == 0 ? 0.0f : f * (r == 1 ? 1e-41f : 1e-37f); }
                        break;
                    case 2:
                        for (auto& f : w) { int r = rand() % 4; if (r == 0) f = 0.0f; else if (r == 1) f = -0.0f; }
                        for (auto& f : x) { int r = rand() % 4; if (r == 0) f = 0.0f; else if (r == 1) f = -0.0f; }
                        if (trial % 2 == 0) { for (auto& f : w) f = (rand() & 1) ? 0.0f : -0.0f; }              // every product a signed zero
                        for (auto& f : c0) f = (rand() & 1) ? 0.0f : -0.0f;
                        break;
                    case 3:
                        for (auto& f : w) if (rand() % 29 == 0) f = sgn() * INFINITY;
                        for (auto& f : x) if (rand() % 97 == 0) f = sgn() * INFINITY;
                        for (auto& f : c0) if (rand() % 7 == 0) f = sgn() * INFINITY;
                        if (trial % 3 == 0) for (auto& f : x) if (rand() % 11 == 0) f = 0.0f;                    // inf * 0
                        break;
                    case 4:
                        for (auto& f : w) if (rand() % 57 == 0) f = bits(0x7f800000u | (urand() & 0x807fffffu) | 1u);  // quiet and signalling, both signs
                        for (auto& f : x) if (rand() % 301 == 0) f = bits(0x7fc00000u | (urand() & 0x803fffffu));
                        for (auto& f : c0) if (rand() % 13 == 0) f = bits(0x7fc00000u | (urand() & 0x803fffffu));
                        break;
                    case 5:
                        for (auto& f : w) f *= (rand() % 3 == 0) ? 1.5e19f : 1e18f;
                        for (auto& f : x) f *= (rand() % 3 == 0) ? 1.5e19f : 1e18f;
                        for (auto& f : c0) f = sgn() * (3.0e38f + 0.4e38f * (float)rand() / RAND_MAX);          // |c| up to FLT_MAX = 3.4028e38
                        break;
                    case 6:
                        for (auto& f : w) f = bits(urand());
                        for (auto& f : x) f = bits(urand());
                        for (auto& f : c0) f = bits(urand());
                        if (trial & 1) { for (auto& f : w) f = bits((urand() & 0x80ffffffu) | ((uint32_t)(100 + rand() % 56) << 23)); // exponents near 1
                                         for (auto& f : x) f = bits((urand() & 0x80ffffffu) | ((uint32_t)(100 + rand() % 56) << 23)); }
                        break;
                    case 7:  // every second step undoes the one before: the accumulator passes through 0 / tiny values again and again
                        for (int k = 0; k + 1 < K; k += 2) {
                            for (int r = 0; r < 4; ++r) w[(k + 1) * 4 + r] = -w[k * 4 + r];
                            for (int l = 0; l < 64; ++l) x[(k + 1) * 64 + l] = x[k * 64 + l];
                        }
                        for (auto& f : c0) { int r = rand() % 4; f = r == 0 ? 0.0f : (r == 1 ? -0.0f : (r == 2 ? f * 1e-39f : f * 1e-7f)); }
                        if (trial & 1) for (auto& f : w) f *= 1e-20f, (void)0;
                        if (trial & 1) for (auto& f : x) f *= 1e-20f;
                        break;
                }
                CHECK(hipMemcpy(dw, w.data(), K * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, x.data(), K * 256, hipMemcpyHostToDevice));
                CHECK(hipMemcpy(dc, c0.data(), 1024, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, dw, dx, dc, K, dm, dv);
                CHECK(hipMemcpy(m.data(), dm, 1024, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(v.data(), dv, 1024, hipMemcpyDeviceToHost));
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        volatile float acc = c0[l * 4 + r];
                        for (int k = 0; k < K; ++k) acc = fmaf(w[k * 4 + r], x[k * 64 + l], acc);
                        h[l * 4 + r] = acc;
                    }
                for (int i = 0; i < 256; ++i) {
                    ++total;
                    uint32_t um, uv, uh;
                    memcpy(&um, &m[i], 4); memcpy(&uv, &v[i], 4); memcpy(&uh, &h[i], 4);
                    const bool nm = m[i] != m[i], nv = v[i] != v[i], nh = h[i] != h[i];
                    if (um != uv) { ++bad_mv; if (!(nm && nv)) ++bad_mv_nn; }
                    if (um != uh) { ++bad_mh; if (!(nm && nh)) ++bad_mh_nn; }
                    n_nan += nh; n_inf += (!nh && (uh & 0x7fffffffu) == 0x7f800000u); n_sub += ((uh & 0x7f800000u) == 0 && (uh & 0x007fffffu) != 0); n_negzero += (uh == 0x80000000u);
                }
            }
            printf("class %d %-52s MFMA vs __fmaf_rn: %ld differ (%ld non-NaN) | MFMA vs host fmaf: %ld differ (%ld non-NaN) of %ld  [results: %ld NaN, %ld inf, %ld subnormal, %ld -0]\n",
                   cl.id, cl.name, bad_mv, bad_mv_nn, bad_mh, bad_mh_nn, total, n_nan, n_inf, n_sub, n_negzero);
        }
    }
    if (getenv("MFMA_PROBE_EXACT_ONLY")) return 0;
    // ---- 3/4. rates ----
    uint64_t* dout; CHECK(hipMalloc(&dout, 16 * 64)); CHECK(hipMemset(dout, 0, 16 * 64));
    uint64_t ho[128];
    auto report = [&](const char* name, int n_instr, int blocks) {
        hipDeviceSynchronize();
        hipMemcpy(ho, dout, 16 * 64, hipMemcpyDeviceToHost);
        double mx = 0;
        for (int b2 = 0; b2 < blocks; ++b2) mx = ho[b2 * 2] > mx ? (double)ho[b2 * 2] : mx;
        printf("%-58s %8.2f cycles / MFMA (wave 0 of %d)\n", name, mx / n_instr, blocks);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate_f32_indep, dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 8 independent accumulators, 1 wave", 512, 1);
        hipLaunchKernelGGL(rate_f32_indep, dim3(1), dim3(256), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 8 independent, 4 waves (1 / SIMD)", 512, 1);
        hipLaunchKernelGGL(rate_f32_indep, dim3(1), dim3(512), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 8 independent, 8 waves (2 / SIMD)", 512, 1);
        hipLaunchKernelGGL((rate_f32_chain<1>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, dependent chain (1 accumulator)", 512, 1);
        hipLaunchKernelGGL((rate_f32_chain<2>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 2 accumulators alternating", 512, 1);
        hipLaunchKernelGGL((rate_f32_chain<3>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 3 accumulators", 510, 1);
        hipLaunchKernelGGL((rate_f32_chain<4>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 16b, 4 accumulators", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<1>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 1 v_fma per MFMA, 1 wave", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<2>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 2 v_fma per MFMA, 1 wave", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<3>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 3 v_fma per MFMA, 1 wave", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<4>), dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 4 v_fma per MFMA, 1 wave", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<2>), dim3(1), dim3(512), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 2 v_fma per MFMA, 8 waves", 512, 1);
        hipLaunchKernelGGL((rate_f32_mixed<4>), dim3(1), dim3(512), 0, 0, dout, 1.5f); if (rep) report("f32 4x4x1 + 4 v_fma per MFMA, 8 waves", 512, 1);
        hipLaunchKernelGGL(rate_valu_only, dim3(1), dim3(64), 0, 0, dout, 1.5f); if (rep) report("v_fma_f32 only (8 chains), 1 wave   [per v_fma]", 512, 1);
        hipLaunchKernelGGL(rate_valu_only, dim3(1), dim3(512), 0, 0, dout, 1.5f); if (rep) report("v_fma_f32 only (8 chains), 8 waves  [per v_fma]", 512, 1);
        hipLaunchKernelGGL(rate_i8_4x4x4, dim3(1), dim3(64), 0, 0, dout, 3); if (rep) report("i8 4x4x4 16b, 8 independent, 1 wave", 512, 1);
        hipLaunchKernelGGL(rate_i8_16x16x64, dim3(1), dim3(64), 0, 0, dout, 3); if (rep) report("i8 16x16x64, 8 independent, 1 wave", 512, 1);
    }
    // ---- i8 4x4x4 layout ----
    {
        int ia[64], ib[64], id[256], *dia, *dib, *did;
        for (int l = 0; l < 64; ++l) {
            // A[i][k] = (i + 1) + 10 k + block ; B[k][j] = 1 << k  (so D[i][j] = sum_k A[i][k] 2^k, independent of j) plus j-dependence via B[0][j] = j + 1
            const int i = l & 3, blk = l >> 2;
            uint32_t av = 0, bv = 0;
            for (int k = 0; k < 4; ++k) av |= (uint32_t)((i + 1 + 10 * k + blk) & 0x7f) << (8 * k);
            for (int k = 0; k < 4; ++k) bv |= (uint32_t)(k == 0 ? (i + 1) : (1 << k)) << (8 * k);
            ia[l] = (int)av; ib[l] = (int)bv;
        }
        CHECK(hipMalloc(&dia, 256)); CHECK(hipMalloc(&dib, 256)); CHECK(hipMalloc(&did, 1024));
        CHECK(hipMemcpy(dia, ia, 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dib, ib, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_i8_kernel, dim3(1), dim3(64), 0, 0, dia, dib, did);
        CHECK(hipMemcpy(id, did, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, j = l & 3;
            int want = 0;
            for (int k = 0; k < 4; ++k) want += (r + 1 + 10 * k + blk) * (k == 0 ? (j + 1) : (1 << k));
            if (id[l * 4 + r] != want) ++bad;
        }
        printf("layout i8 4x4x4: D[lane][r] == sum_k A[4b+r].byte[k] * B[lane].byte[k]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
        if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %d %d %d %d\n", l, id[l * 4], id[l * 4 + 1], id[l * 4 + 2], id[l * 4 + 3]);
    }
    return 0;
}
