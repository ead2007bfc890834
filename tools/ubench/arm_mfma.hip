// VERDICT r01 item 1, measured in isolation: the integer ARM (20 -> 20 -> 20 -> 2 + linear stabiliser, armint.py:180-203) of a
// 16-pixel batch evaluated EXACTLY on the matrix cores with limb-split int8 operands (v_mfma_i32_16x16x64_i8), against the
// int64 definition on the host.  N = the 16 pixels, M = output neurons, K = (input, signed-digit byte).  Products of bytes
// i + j = s share one i32 accumulator tile (|partial| < 2^21); tiles are recombined with shifts in 64 bits.  The accumulator
// layout (lane = pixel n, rows 4 g + r) is the next layer's B-operand layout when tile 1 puts neurons 16..19 at rows
// 0, 4, 8, 12 (and the two stabiliser outputs at rows 1, 2): no cross-lane movement between layers.
// Prints: mismatches against the host (must be 0) and ticks per batch of one lone wave.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/arm_mfma.hip -o tools/ubench/arm_mfma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int NSP = 14, NIF = 6, DIM = 20;

struct Net {  // fixed-point parameters as the decoder sees them (int32 operands: the `narrow` envelope)
    int32_t w1[DIM][DIM], w2[DIM][DIM], w3[DIM][2], ws[DIM][2];   // [in][out]
    int64_t b1[DIM], b2[DIM], b3[2], bs[2];
};

// ---- host reference: armint.py:180-203 in int64 ----
static int64_t g_max_act = 0;  // the scheme carries activations as 3 signed-digit bytes: |act| < 2^23 (128.0 in Q16)
static void ref_arm(const Net& n, const int32_t* v /*[DIM] raw inputs*/, int64_t out[2]) {
    int64_t x[DIM], y[DIM], stab[2];
    for (int i = 0; i < DIM; ++i) x[i] = (int64_t)v[i] << 16;
    for (int o = 0; o < 2; ++o) { int64_t a = n.bs[o]; for (int i = 0; i < DIM; ++i) a += x[i] * n.ws[i][o]; stab[o] = a; }
    for (int o = 0; o < DIM; ++o) { int64_t a = n.b1[o]; for (int i = 0; i < DIM; ++i) a += x[i] * n.w1[i][o]; y[o] = (a < 0 ? 0 : a) >> 16; if (y[o] > g_max_act) g_max_act = y[o]; }
    for (int o = 0; o < DIM; ++o) x[o] = y[o];
    for (int o = 0; o < DIM; ++o) { int64_t a = n.b2[o]; for (int i = 0; i < DIM; ++i) a += x[i] * n.w2[i][o]; y[o] = (a < 0 ? 0 : a) >> 16; if (y[o] > g_max_act) g_max_act = y[o]; }
    for (int o = 0; o < DIM; ++o) x[o] = y[o];
    for (int o = 0; o < 2; ++o) { int64_t a = n.b3[o]; for (int i = 0; i < DIM; ++i) a += x[i] * n.w3[i][o]; out[o] = (a + stab[o]) >> 24; }
}

static inline int limb(int32_t w, int j) {  // signed-digit byte j of w, |w| < 2^23
    if (j < 0 || j > 2) return 0;
    const uint32_t d = ((uint32_t)w + 0x00808080u) ^ 0x00808080u;
    return (int8_t)(d >> (8 * j));
}

// A operands: [mfma][lane] 16 bytes.  mfma order: layer 1: (tile 0, s 0..3), (tile 1, s 0..3); layer 2: (tile 0, s 0..4), (tile 1, s 0..4);
// output: s 0..4.  K-slot kappa = 16 g + beta of lane group g.
struct Tables { i32x4 a[23][64]; int64_t bias1[64][5], bias2[64][5], bias3[64][2], biass[64][2]; };

static int out_of_row(int tile, int m, int* stab) {  // which output a row computes (-1: none); tile 1: neurons at rows 0,4,8,12, stabiliser at rows 1,2
    *stab = -1;
    if (tile == 0) return m;
    if (m % 4 == 0) return 16 + m / 4;
    if (m == 1 || m == 2) { *stab = m - 1; return -1; }
    return -1;
}

static void build_tables(const Net& n, Tables& T) {
    memset(&T, 0, sizeof(T));
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, g = lane >> 4;
        int8_t bytes[16];
        // ---- layer 1 (+ stabiliser rows): K-slots of group 0 = 14 spatial contexts (1 limb), group 1 = 6 IFCE features x 2 limbs
        for (int tile = 0; tile < 2; ++tile)
            for (int s = 0; s < 4; ++s) {
                memset(bytes, 0, 16);
                int st; const int o = out_of_row(tile, m, &st);
                for (int beta = 0; beta < 16; ++beta) {
                    int in = -1, i = 0;
                    if (g == 0 && beta < NSP) { in = beta; i = 0; }
                    if (g == 1 && beta < 2 * NIF) { in = NSP + beta / 2; i = beta % 2; }
                    if (in < 0) continue;
                    if (o >= 0) bytes[beta] = (int8_t)limb(n.w1[in][o], s - i);
                    else if (st >= 0) bytes[beta] = (int8_t)limb(n.ws[in][st], s - i);
                }
                memcpy(&T.a[tile * 4 + s][lane], bytes, 16);
            }
        // ---- layer 2 and output: K-slots of group g = activations {4g, 4g+1, 4g+2, 4g+3, 16+g} x 3 limbs
        for (int tile = 0; tile < 2; ++tile)
            for (int s = 0; s < 5; ++s) {
                memset(bytes, 0, 16);
                int st; const int o = out_of_row(tile, m, &st);
                for (int beta = 0; beta < 15; ++beta) {
                    const int a = beta / 3, i = beta % 3, in = a < 4 ? 4 * g + a : 16 + g;
                    if (o >= 0) bytes[beta] = (int8_t)limb(n.w2[in][o], s - i);
                }
                memcpy(&T.a[8 + tile * 5 + s][lane], bytes, 16);
            }
        for (int s = 0; s < 5; ++s) {
            memset(bytes, 0, 16);
            for (int beta = 0; beta < 15; ++beta) {
                const int a = beta / 3, i = beta % 3, in = a < 4 ? 4 * g + a : 16 + g;
                if (m < 2) bytes[beta] = (int8_t)limb(n.w3[in][m], s - i);
            }
            memcpy(&T.a[18 + s][lane], bytes, 16);
        }
        // biases of the rows this lane's accumulators hold: tile 0 rows 4g+r (r = 0..3), tile 1 row 4g (neuron 16+g)
        for (int r = 0; r < 4; ++r) { T.bias1[lane][r] = n.b1[4 * g + r]; T.bias2[lane][r] = n.b2[4 * g + r]; }
        T.bias1[lane][4] = n.b1[16 + g]; T.bias2[lane][4] = n.b2[16 + g];
        for (int o = 0; o < 2; ++o) { T.bias3[lane][o] = n.b3[o]; T.biass[lane][o] = n.bs[o]; }
    }
}


// ---- host emulation of the kernel's arithmetic (same tables, same packing), to separate scheme errors from layout errors ----
static void emu_mfma(const i32x4* a /*[64]*/, const int8_t (*b)[16] /*[64 lanes][16]*/, int32_t d[16][16] /*[row][col]*/) {
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            int32_t acc = 0;
            for (int g = 0; g < 4; ++g) {
                int8_t ab[16]; memcpy(ab, &a[16 * g + m], 16);
                for (int k = 0; k < 16; ++k) acc += (int32_t)ab[k] * b[16 * g + n][k];
            }
            d[m][n] = acc;
        }
}
static void emu_pack15(const int32_t a[5], int8_t out[16]) {
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 3; ++j) out[3 * i + j] = (int8_t)limb(a[i], j);
    out[15] = 0;
}
static void emu_batch(const Net& n, const Tables& T, const int32_t* v /*[16][DIM]*/, int64_t* out /*[16][2]*/) {
    static int8_t b[64][16]; static int32_t c[23][16][16];
    memset(b, 0, sizeof(b));
    for (int px = 0; px < 16; ++px) {
        for (int k = 0; k < NSP; ++k) b[px][k] = (int8_t)v[px * DIM + k];
        for (int f = 0; f < NIF; ++f) { const uint32_t d = ((uint32_t)v[px * DIM + NSP + f] + 0x8080u) ^ 0x8080u; b[16 + px][2 * f] = (int8_t)d; b[16 + px][2 * f + 1] = (int8_t)(d >> 8); }
    }
    for (int i = 0; i < 8; ++i) emu_mfma(T.a[i], b, c[i]);
    auto comb = [&](int first, int ns, int row, int px, int64_t bias) { int64_t a = 0; for (int s = 0; s < ns; ++s) a += (int64_t)c[first + s][row][px] << (8 * s); return bias + (ns == 4 ? a << 16 : a); };  // only the raw inputs of layer 1 carry the << 16
    int32_t act[16][DIM]; int64_t stab[16][2];
    for (int px = 0; px < 16; ++px) {
        for (int o = 0; o < 16; ++o) { const int64_t p = comb(0, 4, o, px, n.b1[o]); act[px][o] = (int32_t)((p < 0 ? 0 : p) >> 16); }
        for (int o = 16; o < 20; ++o) { const int64_t p = comb(4, 4, 4 * (o - 16), px, n.b1[o]); act[px][o] = (int32_t)((p < 0 ? 0 : p) >> 16); }
        for (int o = 0; o < 2; ++o) stab[px][o] = comb(4, 4, 1 + o, px, n.bs[o]);
    }
    for (int layer = 0; layer < 2; ++layer) {
        for (int px = 0; px < 16; ++px) for (int g = 0; g < 4; ++g) { const int32_t a5[5] = {act[px][4 * g], act[px][4 * g + 1], act[px][4 * g + 2], act[px][4 * g + 3], act[px][16 + g]}; emu_pack15(a5, b[16 * g + px]); }
        if (layer == 0) {
            for (int i = 8; i < 18; ++i) emu_mfma(T.a[i], b, c[i]);
            for (int px = 0; px < 16; ++px) {
                int32_t nx[DIM];
                for (int o = 0; o < 16; ++o) { const int64_t p = comb(8, 5, o, px, n.b2[o]); nx[o] = (int32_t)((p < 0 ? 0 : p) >> 16); }
                for (int o = 16; o < 20; ++o) { const int64_t p = comb(13, 5, 4 * (o - 16), px, n.b2[o]); nx[o] = (int32_t)((p < 0 ? 0 : p) >> 16); }
                memcpy(act[px], nx, sizeof(nx));
            }
        } else {
            for (int i = 18; i < 23; ++i) emu_mfma(T.a[i], b, c[i]);
            for (int px = 0; px < 16; ++px) for (int o = 0; o < 2; ++o) out[px * 2 + o] = (comb(18, 5, o, px, n.b3[o]) + stab[px][o]) >> 24;
        }
    }
}

__device__ __forceinline__ int32_t relu_shift(int64_t pre) { return static_cast<int32_t>((pre < 0 ? 0 : pre) >> 16); }
__device__ __forceinline__ uint32_t digits(int32_t a) { return (static_cast<uint32_t>(a) + 0x00808080u) ^ 0x00808080u; }  // bytes 0..2 = signed digits

// 5 activations -> 15 signed-digit bytes in 4 dwords (byte 15 = 0)
__device__ __forceinline__ i32x4 pack15(const int32_t (&a)[5]) {
    uint32_t d[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = digits(a[i]) & 0x00ffffffu;
    i32x4 r;
    r[0] = static_cast<int>(d[0] | (d[1] << 24));
    r[1] = static_cast<int>((d[1] >> 8) | (d[2] << 16));
    r[2] = static_cast<int>((d[2] >> 16) | (d[3] << 8));
    r[3] = static_cast<int>(d[4]);
    return r;
}

// ctx: [batch][16 pixels][DIM] raw inputs.  out: [batch][16][2]
__global__ __launch_bounds__(64) void arm_mfma_kernel(const Tables* __restrict__ T, const int32_t* __restrict__ ctx, int64_t* __restrict__ out, int n_batches,
                                                       uint64_t* ticks) {
    __shared__ i32x4 s_a[23][64];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    for (int i = lane; i < 23 * 64; i += 64) s_a[i / 64][i % 64] = T->a[i / 64][i % 64];
    int64_t b1[5], b2[5], b3[2], bs[2];
#pragma unroll
    for (int r = 0; r < 5; ++r) { b1[r] = T->bias1[lane][r]; b2[r] = T->bias2[lane][r]; }
#pragma unroll
    for (int o = 0; o < 2; ++o) { b3[o] = T->bias3[lane][o]; bs[o] = T->biass[lane][o]; }
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int bt = 0; bt < n_batches; ++bt) {
        const int32_t* v = ctx + (static_cast<size_t>(bt) * 16 + n) * DIM;
        // ---- B operand of layer 1: group 0 = 14 context bytes, group 1 = 6 features x 2 signed-digit bytes
        i32x4 bop = {0, 0, 0, 0};
        if (g == 0) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NSP; ++k) w[k / 4] |= (static_cast<uint32_t>(v[k]) & 0xffu) << (8 * (k % 4));
            bop = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        } else if (g == 1) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int f = 0; f < NIF; ++f) {
                const uint32_t d = (static_cast<uint32_t>(v[NSP + f]) + 0x00008080u) ^ 0x00008080u;  // 2 signed digits (|feature| < 2^15)
                w[(2 * f) / 4] |= (d & 0xffffu) << (8 * ((2 * f) % 4));
            }
            bop = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
        }
        // ---- layer 1 + stabiliser: 2 tiles x 4 limb sums
        i32x4 c[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) c[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(s_a[t * 4 + s][lane], bop, i32x4{0, 0, 0, 0}, 0, 0, 0);
        auto comb4 = [](const i32x4 (&cc)[4], int r, int64_t bias) {  // bias + (sum_s c_s 2^(8 s)) 2^16
            const int32_t lo = cc[0][r] + (cc[1][r] << 8), hi = cc[2][r] + (cc[3][r] << 8);
            return bias + (static_cast<int64_t>(lo) << 16) + (static_cast<int64_t>(hi) << 32);
        };
        int32_t act[5];
#pragma unroll
        for (int r = 0; r < 4; ++r) act[r] = relu_shift(comb4(c[0], r, b1[r]));
        act[4] = relu_shift(comb4(c[1], 0, b1[4]));
        const int64_t stab0 = comb4(c[1], 1, bs[0]), stab1 = comb4(c[1], 2, bs[1]);  // meaningful in group 0 (rows 1, 2 of tile 1)
        // ---- layer 2: 2 tiles x 5 limb sums
        i32x4 bop2 = pack15(act);
        i32x4 d[2][5];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 5; ++s) d[t][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(s_a[8 + t * 5 + s][lane], bop2, i32x4{0, 0, 0, 0}, 0, 0, 0);
        auto comb5 = [](const i32x4 (&cc)[5], int r, int64_t bias) {
            const int32_t lo = cc[0][r] + (cc[1][r] << 8), mid = cc[2][r] + (cc[3][r] << 8);
            return bias + lo + (static_cast<int64_t>(mid) << 16) + (static_cast<int64_t>(cc[4][r]) << 32);  // Q16 x Q16: no input shift after layer 1
        };
#pragma unroll
        for (int r = 0; r < 4; ++r) act[r] = relu_shift(comb5(d[0], r, b2[r]));
        act[4] = relu_shift(comb5(d[1], 0, b2[4]));
        // ---- output layer: 1 tile x 5 limb sums, rows 0 (mu) and 1 (log-scale) live in group 0
        i32x4 bop3 = pack15(act);
        i32x4 e[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) e[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(s_a[18 + s][lane], bop3, i32x4{0, 0, 0, 0}, 0, 0, 0);
        if (g == 0) {
            out[(static_cast<size_t>(bt) * 16 + n) * 2 + 0] = (comb5(e, 0, b3[0]) + stab0) >> 24;
            out[(static_cast<size_t>(bt) * 16 + n) * 2 + 1] = (comb5(e, 1, b3[1]) + stab1) >> 24;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[0] = t1 - t0;
}

int main() {
    srand(7);
    Net net;
    auto rw = [](int bits) { return (int32_t)((rand() % (2 << bits)) - (1 << bits)); };
    for (int i = 0; i < DIM; ++i)
        for (int o = 0; o < DIM; ++o) {
            net.w1[i][o] = rw(i < NSP ? 12 : 4) + (i == o ? (i < NSP ? 65536 : 256) : 0);  // residual folded in (armint.py:114-124)
            net.w2[i][o] = rw(12) + (i == o ? 65536 : 0);
        }
    for (int i = 0; i < DIM; ++i) for (int o = 0; o < 2; ++o) { net.w3[i][o] = rw(14); net.ws[i][o] = rw(i < NSP ? 14 : 6); }
    auto rb = []() { return ((int64_t)rand() << 3) - ((int64_t)RAND_MAX << 2); };
    for (int o = 0; o < DIM; ++o) { net.b1[o] = rb(); net.b2[o] = rb(); }
    for (int o = 0; o < 2; ++o) { net.b3[o] = rb(); net.bs[o] = rb(); }
    const int NB = 4096;
    std::vector<int32_t> ctx((size_t)NB * 16 * DIM);
    for (size_t p = 0; p < (size_t)NB * 16; ++p) {
        for (int k = 0; k < NSP; ++k) ctx[p * DIM + k] = (rand() % 3 == 0) ? (rand() % 128) - 64 : (rand() % 9) - 4;
        for (int f = 0; f < NIF; ++f) ctx[p * DIM + NSP + f] = (rand() % 8192) - 4096;
    }
    std::vector<int64_t> want((size_t)NB * 16 * 2), got((size_t)NB * 16 * 2);
    for (size_t p = 0; p < (size_t)NB * 16; ++p) ref_arm(net, &ctx[p * DIM], &want[p * 2]);
    printf("largest hidden activation of the test set: %lld (limit of the 3-byte carry: %d)\n", (long long)g_max_act, 1 << 23);
    Tables* hT = new Tables; build_tables(net, *hT);
    {   // the scheme itself, on the host
        size_t ebad = 0; int64_t eo[32];
        for (int bt = 0; bt < 64; ++bt) { emu_batch(net, *hT, &ctx[(size_t)bt * 16 * DIM], eo); for (int i = 0; i < 32; ++i) ebad += eo[i] != want[(size_t)bt * 32 + i]; }
        printf("host emulation of the limb scheme vs int64 reference: %zu / %d outputs differ\n", ebad, 64 * 32);
    }
    Tables* dT; int32_t* dctx; int64_t* dout; uint64_t* dt;
    CHECK(hipMalloc(&dT, sizeof(Tables))); CHECK(hipMalloc(&dctx, ctx.size() * 4)); CHECK(hipMalloc(&dout, got.size() * 8)); CHECK(hipMalloc(&dt, 64));
    CHECK(hipMemcpy(dT, hT, sizeof(Tables), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dctx, ctx.data(), ctx.size() * 4, hipMemcpyHostToDevice));
    uint64_t ticks = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(arm_mfma_kernel, dim3(1), dim3(64), 0, 0, dT, dctx, dout, NB, dt);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) if (got[i] != want[i]) { if (bad < 5) printf("  mismatch %zu: got %lld want %lld\n", i, (long long)got[i], (long long)want[i]); ++bad; }
    printf("limb-split int8 MFMA ARM vs int64 reference: %zu / %zu outputs differ\n", bad, got.size());
    printf("one lone wave: %.0f ticks per 16-pixel batch (23 MFMAs 16x16x64 + limb recombination; inputs from global memory)\n", (double)ticks / NB);
    printf("production producers (level-3 profile, DESIGN.md 7.1): 2 x 5772 ticks of MLP per 16 pixels\n");
    return bad != 0;
}
