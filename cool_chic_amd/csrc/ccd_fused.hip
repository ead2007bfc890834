// ccd_fused.hip - the float path of a cool-chic in ONE kernel: int8 latent pyramid -> learned x2 upsampling of every
// level -> synthesis conv stack -> float samples and / or integer samples.  Only the int8 latent grids are read from
// HBM and only the output is written (algorithmic traffic S + C bytes per pixel for 8-bit integer output, SURVEY.md
// section 8d: 4.33 B/px at Kodak-HOP); the dense [L][H][W] f32 stack of the unfused path never exists.
//
// Reference behaviour (paths relative to /root/reference/coolchic):
//   component/core/upsampling.py:463-500,287-330,158-203   Upsampling.forward (training-mode 2-D kron kernels)
//   component/core/synthesis.py:61-76,272-294               Synthesis.forward
//   bitstream/decode.py:191-206, io/format/png.py:57        rounding / clamping to integer samples
//
// Numerics contract (unchanged from ccd_float.hip / ccd_synth_fused.hip): every output is the oracle's fmaf chain in
// the oracle's order (oracle/cc_oracle.c sections 8-9).  ALL chains run on the matrix cores:
// v_mfma_f32_4x4x1_16b_f32 computes D[i] = fma(A[i], B, C[i]) per lane with ONE rounding per product and accumulates in
// issue order (tools/ubench/mfma_probe.hip: 0 of 51 200 words differ from the __fmaf_rn chain), so a chain of such
// instructions IS the fmaf chain, four outputs at a time.  The file is compiled with -ffp-contract=off.
//
// Why the matrix cores for a ~1 kMAC/pixel network: a wave issues one instruction every ~5-7 cycles here whatever its
// type, so the kernel is bound by its INSTRUCTION COUNT; one 4x4x1 MFMA step replaces four v_fma of four different
// outputs, and its weights come from a register that holds 16 steps (CBSZ / ABID broadcast), not from loads.
//
// Structure.  A 256-thread workgroup owns a 64 x 32 "extended" tile of the finest level (interior + an even halo margin
// >= the number of 3x3 layers) and walks the pyramid coarse -> fine inside LDS.  Level i's footprint is rows / columns
// [q0 - 2, q1 + 2] of the quads (2 x 2 outputs) the level above needs; footprints shrink by half per level and stop at
// ~10 x 10, so recomputing the coarse levels per tile costs a few % of the tile.  Every LDS buffer has a compile-time
// pitch and holds its footprint UNCLIPPED: positions outside the grid carry what the reference's padding gives them
// (replicate for the x2 filters and the 3x3 layers, zero for the 7x7 filter's input), written by the producer.  So a
// consumer's window is ONE base address plus immediate offsets - no clamps, no per-sample address arithmetic:
//   S1  the int8 latents of every level's footprint (+ 4) -> f32 tiles in LDS, zero outside the grid;
//   S2  phase A: channel i of level i = the level's own latent through the 7x7 pre-concatenation filter (the plain latent at
//       the coarsest level), all levels at once; phase B: level i = L-2 .. 1, channels > i from level i + 1 (x2 filter).
//       A lane owns a 2 x 2 output quad = the 4 rows of the MFMA, one step per sample of its source window;
//   S3  level 0: the lane's L dense values per pixel stay in registers and feed the first 1x1 layer (hidden units in tiles
//       of 4 = the 4 rows), ReLU, second 1x1 layer and the stabiliser, all in registers (4 "sets" of 64 pixels per wave);
//   S4  the 3x3 layers ping-pong between two LDS tiles; each tap is one step whose B operand is a register of the lane's
//       4 x 4 window; the last layer continues into the stabiliser add, the output transform and the stores.
// A quad outside the grid is evaluated at the clamped quad and each of its positions takes the output of the clamped
// position (replicate), only in tiles that touch the border.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ccd_device.hpp"

namespace ccd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFdThreads = 256;
constexpr int kFdEW = 64, kFdEH = 32;            // extended tile of the finest level
constexpr int kFdTP = 68, kFdTRows = 34;         // 3x3-layer tiles: pitch / rows incl. one guard row + column on every side
constexpr int kFdTileCh = kFdTP * kFdTRows;      // words per channel

// Largest footprint of level i for a 64 x 32 tile whose origin is even (rows, cols): 32 -> 16 quads + 4 = 20 -> <= 11 + 4 ...
__host__ __device__ constexpr int fd_reg_h(int i) { return i == 0 ? 32 : i == 1 ? 20 : i == 2 ? 15 : i == 3 ? 12 : i == 4 ? 11 : 10; }
__host__ __device__ constexpr int fd_reg_w(int i) { return i == 0 ? 64 : i == 1 ? 36 : i == 2 ? 23 : i == 3 ? 16 : i == 4 ? 13 : i == 5 ? 11 : 10; }
__host__ __device__ constexpr int fd_pl(int i) { return fd_reg_h(i) * fd_reg_w(i); }  // plane of a level's channel (pitch fd_reg_w)
// latent tiles: footprint + 4 on every side (3 for the 7x7 filter, 1 because a quad may start one row above the footprint),
// origin column moved left to an odd coordinate so that 7x7 windows start 8-byte aligned; pitch a multiple of 4 (S1 stores
// four samples at a time)
__host__ __device__ constexpr int fd_lat_h(int i) { return fd_reg_h(i) + 8; }
__host__ __device__ constexpr int fd_lat_p(int i) { return (fd_reg_w(i) + 9 + 3) & ~3; }
__host__ __device__ constexpr int fd_lat_elems(int i) { return fd_lat_h(i) * fd_lat_p(i); }
// most quads of level i (S2 wave-items of 64 quads)
__host__ __device__ constexpr int fd_max_q(int i) { return (fd_reg_h(i) / 2 + 1) * (fd_reg_w(i) / 2 + 1); }

// LDS layout in 4-byte words; everything but the total is a compile-time function of (levels, channels)
struct FdLayout { int k2, lat0, va, pc1, rest, vb, tile_a, tile_b, par; };
__host__ __device__ constexpr int fd_lat_off_rel(int i) { int n = 0; for (int j = 1; j < i; ++j) n += fd_lat_elems(j); return n; }  // levels >= 1, from `rest`
__host__ __device__ constexpr int fd_pc_off_rel(int cin, int i) { int n = fd_lat_off_rel(cin); for (int j = 2; j < i; ++j) n += fd_pl(j); return n; }  // levels >= 2
__host__ __device__ constexpr FdLayout fd_layout(int cin, int c) {
    FdLayout L{};
    int o = 0;
    L.k2 = o; o += 24 * cin;                                  // [level][x2 | 7x7][12]: 10 kron products, then zeros
    L.lat0 = o; o += fd_lat_elems(0);
    L.va = o; o += (cin > 2 ? cin - 2 : 0) * fd_pl(1);       // levels 1, 3, ..: channels level + 1 .. L - 1
    L.pc1 = o; o += fd_pl(1);
    L.rest = o;                                               // latent tiles of levels >= 1, own channels of levels >= 2, stack B
    L.vb = o + fd_pc_off_rel(cin, cin);
    o = L.vb + (cin > 3 ? cin - 3 : 0) * fd_pl(2);           // levels 2, 4, ..
    // 3x3-layer tiles alias the pyramid: A is written at the end of S3 (only lat0 / va / pc1 are still read), B in S4
    const int tile = c * kFdTileCh;
    L.tile_b = L.lat0; L.tile_a = L.rest;
    if (L.rest - L.lat0 < tile) L.tile_a = L.lat0 + tile;
    if (L.tile_a + tile > o) o = L.tile_a + tile;
    L.par = (o + 3) & ~3;
    return L;
}

extern __shared__ __attribute__((aligned(16))) float fd_smem[];

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_down(F&& f) {  // I, I - 1, .., N
    if constexpr (I >= N) {
        f(std::integral_constant<int, I>{});
        static_for_down<I - 1, N>(f);
    }
}

// One multiply-add step on the matrix cores: D[i] = fma(W_Q[i], b, c[i]) for the lane's pixel, W_Q = the 4 weights of
// step Q, held in lanes 4 (Q % 16) .. + 3 of register Q / 16 of `w`.
template <int Q, int NW>
__device__ __forceinline__ f32x4 mstep(const float (&w)[NW], float b, f32x4 c) {
    static_assert(Q / 16 < NW, "weight register out of range");
    return __builtin_amdgcn_mfma_f32_4x4x1f32(w[Q / 16], b, c, 4, Q % 16, 0);
}

__device__ __forceinline__ int fd_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float fd_relu(float a) { return a > 0.0f ? a : 0.0f; }

// decode.py:191-206 + png.py:57 / yuv.py:152-160: q = round(maxv x) / maxv, clamp to [0, 1], round(q maxv) / maxv again, then
// the writer's round(q maxv).  With r = clamp(rint(maxv x), 0, maxv) an integer <= 65535, q = RN(r / maxv) and
// RN(q maxv) is within r 2^-23 < 0.5 of r, so every later rounding returns r: the whole chain is rint, clamp.
// (tests/test_oracle_golden.py::test_quantise_shortcut checks it against the oracle's literal chain.)
__device__ __forceinline__ unsigned fd_quantise(float x, float maxv) {
    float r = rintf(maxv * x);
    r = r < 0.0f ? 0.0f : (r > maxv ? maxv : r);
    return static_cast<unsigned>(r);
}

// The upsampling filters as MFMA steps: the 4 rows are the 4 outputs of the lane's quad (row = 2 dy + dx), one step per
// sample of the lane's source window.  A sample that an output does not use has weight 0 for that row:
// fma(v, 0, acc) == acc bit for bit (v is finite, acc is never -0), so every output sees exactly its own taps in its own order.
//
// x2 transposed conv (k = 8, replicate pad 4, crop 11): 5 x 5 window v[a][b] = source (qy - 2 + a, qx - 2 + b).  Output row
// 2 qy uses ky = 1, 3, 5, 7 on window rows 3, 2, 1, 0, output row 2 qy + 1 uses ky = 0, 2, 4, 6 on rows 4, 3, 2, 1 (same
// for columns): taps in ky, kx ascending order = window rows and columns DESCENDING.  Step s <-> (a, b) = (4 - s / 5, 4 - s % 5).
// Pre-concatenation 7x7 conv (zero padding, + residual): 8 x 8 window v[a][b] = latent (2 qy - 3 + a, 2 qx - 3 + b), zeros
// outside the grid (a zero sample contributes nothing, which is the oracle's skipping of those taps).  Output (dy, dx)
// uses tap (ky, kx) = (a - dy, b - dx); step s <-> (a, b) = (s / 8, s % 8), ascending.
//
// The weight of (lane, step) is one of the 10 kron products of the level's filter (k2_index) or zero; WHICH one depends on
// the lane and the step only: 4-bit table indices for the 2 + 4 weight registers, packed once per kernel.
__device__ __forceinline__ uint32_t fd_weight_indices(int lane) {
    const int ph = lane & 3, dy = ph >> 1, dx = ph & 1;
    uint32_t pack = 0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int st = 16 * v + (lane >> 2);
        const int a = 4 - st / 5, b = 4 - st % 5;
        const int ky = dy == 0 ? 2 * (3 - a) + 1 : 2 * (4 - a), kx = dx == 0 ? 2 * (3 - b) + 1 : 2 * (4 - b);
        const bool ok = st < 25 && ky >= 0 && ky < 8 && kx >= 0 && kx < 8;
        const int fy = ky < 4 ? ky : 7 - ky, fx = kx < 4 ? kx : 7 - kx;  // symmetric filter (a b c d d c b a), upsampling.py:42-64
        pack |= static_cast<uint32_t>(ok ? k2_index(fy & 3, fx & 3) : 10) << (4 * v);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int st = 16 * v + (lane >> 2);
        const int ky = st / 8 - dy, kx = st % 8 - dx;
        const bool ok = ky >= 0 && ky < 7 && kx >= 0 && kx < 7;
        const int fy = ky < 4 ? ky : 6 - ky, fx = kx < 4 ? kx : 6 - kx;  // symmetric filter (a b c d c b a)
        pack |= static_cast<uint32_t>(ok ? k2_index(fy & 3, fx & 3) : 10) << (4 * (2 + v));
    }
    return pack;
}
__device__ __forceinline__ void fd_tconv_weights(const float* k2 /*12-entry table, LDS*/, uint32_t pack, float (&wt)[2]) {
#pragma unroll
    for (int v = 0; v < 2; ++v) wt[v] = k2[(pack >> (4 * v)) & 15u];
}
__device__ __forceinline__ void fd_preconv_weights(const float* k2, uint32_t pack, float (&wt)[4]) {
#pragma unroll
    for (int v = 0; v < 4; ++v) wt[v] = k2[(pack >> (4 * (2 + v))) & 15u];
}

// 5 x 5 window at `base` (pitch P words) -> the quad's 4 outputs
template <int P>
__device__ __forceinline__ f32x4 fd_tconv_quad(const float* base, const float (&wt)[2]) {
    float v[5][5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) v[a][b] = base[a * P + b];
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<0, 25>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        acc = mstep<st>(wt, v[4 - st / 5][4 - st % 5], acc);
    });
    return acc;
}
// two independent quads interleaved: the 13-cycle dependent-accumulator latency hides behind the 9-cycle issue interval
template <int P>
__device__ __forceinline__ void fd_tconv_quad2(const float* base0, const float* base1, const float (&wt)[2], f32x4& r0, f32x4& r1) {
    float v0[5][5], v1[5][5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) { v0[a][b] = base0[a * P + b]; v1[a][b] = base1[a * P + b]; }
    // all reads of the two windows first: left alone, the scheduler (short of registers) issues each read one step before its
    // MFMA and the two chains run at LDS latency instead of MFMA latency.  (The single-chain variants above and below are bound
    // by the dependent-MFMA latency either way; grouping their reads made S2 10 % slower.)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 a0 = {0.0f, 0.0f, 0.0f, 0.0f}, a1 = a0;
    static_for<0, 25>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        a0 = mstep<st>(wt, v0[4 - st / 5][4 - st % 5], a0);
        a1 = mstep<st>(wt, v1[4 - st / 5][4 - st % 5], a1);
    });
    r0 = a0; r1 = a1;
}
// 8 x 8 window at `base` (8-byte aligned, even pitch P) -> the quad's 4 outputs incl. the residual
template <int P>
__device__ __forceinline__ f32x4 fd_preconv_quad(const float* base, const float (&wt)[4]) {
    float v[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(base + a * P + 2 * b);
            v[a][2 * b] = t[0]; v[a][2 * b + 1] = t[1];
        }
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<0, 64>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        acc = mstep<st>(wt, v[st / 8][st % 8], acc);
    });
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[ph] = acc[ph] + v[(ph >> 1) + 3][(ph & 1) + 3];
    return acc;
}
// output of the (clamped) position (cy, cx) among the 4 outputs of the quad that holds it
__device__ __forceinline__ float fd_pick(f32x4 r, int cy, int cx) {
    const int i = ((cy & 1) << 1) | (cx & 1);
    return i == 0 ? r[0] : (i == 1 ? r[1] : (i == 2 ? r[2] : r[3]));
}

// S1: a thread fetches FOUR neighbouring samples of a level's latent tile with one (unaligned) dword load: item t of level i
// is row t / (pitch / 4), columns 4 (t % (pitch / 4)) .. + 3.
__host__ __device__ constexpr int fd_s1_items(int i) { return fd_lat_h(i) * (fd_lat_p(i) / 4); }
__host__ __device__ constexpr int fd_s1_rounds(int i) { return (fd_s1_items(i) + kFdThreads - 1) / kFdThreads; }
__host__ __device__ constexpr int fd_s1_slot(int i) { int n = 0; for (int j = 0; j < i; ++j) n += fd_s1_rounds(j); return n; }
// S2 wave-items (64 quads / samples) of level i in phase A and per channel in phase B, and the wave that takes the first one
__host__ __device__ constexpr int fd_wi_a(int cin, int i) { return ((i == cin - 1 ? fd_pl(i) : fd_max_q(i)) + 63) / 64; }
__host__ __device__ constexpr int fd_rot_a(int cin, int i) { int n = 0; for (int j = 1; j < i; ++j) n += fd_wi_a(cin, j); return n & 3; }
__host__ __device__ constexpr int fd_wi_q(int i) { return (fd_max_q(i) + 63) / 64; }

struct FdWork { int32_t frame, tile_first, tile_count, pad; };

// -DCCD_FD_PROFILE: cycles of wave 0 of every workgroup per phase, summed over the launch (ccd_debug_fd_profile)
#ifdef CCD_FD_PROFILE
__device__ unsigned long long fd_prof[16];
#define FDP_T() __builtin_amdgcn_s_memtime()
#define FDP_ADD(slot, t0) do { if (tid == 0) atomicAdd(&fd_prof[slot], __builtin_amdgcn_s_memtime() - (t0)); } while (0)
#else
#define FDP_T() 0ull
#define FDP_ADD(slot, t0) (void)(t0)
#endif

// CIN = latent levels = input channels of the synthesis, C = its output channels (both fix register arrays, LDS layout
// and the immediate operands of the MFMA steps).
template <int CIN, int C>
__global__ __launch_bounds__(kFdThreads, 2) void decode_fused_kernel(const FusedDec* __restrict__ frames, const FdWork* __restrict__ work) {
    static_assert(CIN >= 2 && CIN <= kFdMaxLevels, "levels");
    constexpr int CT = (C + 3) / 4;                  // output-channel tiles of 4
    constexpr int NWV = (CIN + 4 * CT + 15) / 16;    // weight registers per hidden tile: CIN first-layer steps + 4 CT second-layer steps
    constexpr int NWS = (CIN * CT + 15) / 16;        // stabiliser
    constexpr int NWC = (9 * C * CT + 15) / 16;      // one 3x3 layer
    constexpr int NWO = (C * CT + 15) / 16;          // output transform
    constexpr FdLayout L = fd_layout(CIN, C);
    typedef const float __attribute__((address_space(1)))* gcf_t;
    typedef const int8_t __attribute__((address_space(1)))* gci8_t;

    const FdWork wk = work[blockIdx.x];
    const FusedDec& p = frames[wk.frame];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.h, W = p.w;
    float* const s_k2 = fd_smem + L.k2;
    float* const s_par = fd_smem + L.par;

    // ---- once per workgroup: parameters (MFMA order, see FusedDec) and filter tables -> LDS ----------------------------
    const unsigned long long tp0 = FDP_T();
    {
        const gcf_t src = (gcf_t)p.params;
        for (int i = tid * 4; i < p.n_params; i += kFdThreads * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(src + i);
            *reinterpret_cast<f32x4*>(s_par + i) = v;
        }
        for (int i = tid; i < CIN * 24; i += kFdThreads) {
            const int lvl = i / 24, j = i - lvl * 24;
            s_k2[i] = j < 10 ? p.k2u[lvl][j] : (j >= 12 && j < 22 ? p.k2p[lvl][j - 12] : 0.0f);
        }
    }
    const uint32_t wpack = fd_weight_indices(lane);
    FDP_ADD(0, tp0);
    const int RM = p.margin;
    const int iw = kFdEW - 2 * RM, ih = kFdEH - 2 * RM;
    const int bitdepth = p.bitdepth, write_planes = p.write_planes, has_stab = p.has_stab, n_conv = p.n_conv;
    const int relu0 = p.relu0, relu1 = p.relu1, n_ht = p.n_tiles_hidden;
    const int wq_off = p.wq_off, b0_off = p.b0_off, b1_off = p.b1_off, stab_off = p.stab_off, stabb_off = p.stabb_off;
    const int out_off = p.out_off, outb_off = p.outb_off;
    float* const out_f32 = p.out;
    void* const plane_ptr[3] = {p.plane[0], p.plane[1], p.plane[2]};
    const float maxv = static_cast<float>((1 << bitdepth) - 1);
    const int qxl = lane & 31, qyl = lane >> 5;   // S3 / S4: a wave-pass is 32 x 2 quads (64 x 4 pixels)

    for (int tile = wk.tile_first; tile < wk.tile_first + wk.tile_count; ++tile) {
        const int by = tile / p.tiles_x, bx = tile - by * p.tiles_x;
        const int tx0 = bx * iw - RM, ty0 = by * ih - RM;  // image coordinate of extended (0, 0); both even
        const unsigned long long tp1 = FDP_T();
        // S1 / S2 are chains of short dependent steps; S3 / S4 are long MFMA streams.  The two workgroups of a CU share each
        // SIMD's issue slots and matrix pipe "by priority, then age": the latency-bound phases go first.
        __builtin_amdgcn_s_setprio(2);
        // ---- footprints (wave-uniform scalars, static indices): level 0 = the extended tile, level i + 1 = [q0 - 2, q1 + 2]
        int ay[CIN], ax[CIN], fh[CIN], fw[CIN], gh[CIN], gw[CIN];
        bool bord[CIN];  // the footprint leaves the grid: replicate handling needed at this level
        {
            int y0 = ty0, y1 = ty0 + kFdEH - 1, x0 = tx0, x1 = tx0 + kFdEW - 1;
            static_for<0, CIN>([&](auto ll) {
                constexpr int i = decltype(ll)::value;
                ay[i] = y0; ax[i] = x0; fh[i] = y1 - y0 + 1; fw[i] = x1 - x0 + 1;
                gh[i] = (H + (1 << i) - 1) >> i; gw[i] = (W + (1 << i) - 1) >> i;  // grid i = ceil(grid 0 / 2^i)
                bord[i] = y0 < 0 || x0 < 0 || y1 > gh[i] - 1 || x1 > gw[i] - 1;
                y0 = (y0 >> 1) - 2; y1 = (y1 >> 1) + 2; x0 = (x0 >> 1) - 2; x1 = (x1 >> 1) + 2;
            });
        }
        // ---- S1: the latent bytes of every level's tile are requested back-to-back, four samples per (unaligned) dword load from
        // a clamped address (a conditional load makes the compiler wait at every join), and stored as f32 after ONE wait;
        // zero outside the grid.  A dword that would cross the row's right end starts at gw - 4 and is shifted instead.
        uint32_t ld_val[fd_s1_slot(CIN)];
        typedef const uint32_t __attribute__((address_space(1), aligned(1)))* gcu32_t;
        static_for<0, CIN>([&](auto ll) {
            constexpr int i = decltype(ll)::value;
            constexpr int DW = fd_lat_p(i) / 4;
            const int oy = ay[i] - 4, ox = (ax[i] - 5) | 1;
            const gci8_t src = (gci8_t)p.lat[i];
            static_for<0, fd_s1_rounds(i)>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const int t = min(tid + k * kFdThreads, fd_s1_items(i) - 1);
                const int r = t / DW, j = t - r * DW;
                const int cy = fd_clamp(oy + r, 0, gh[i] - 1), xs = fd_clamp(ox + 4 * j, 0, max(gw[i] - 4, 0));
                ld_val[fd_s1_slot(i) + k] = *(gcu32_t)(src + static_cast<uint32_t>(cy * gw[i] + xs));
            });
        });
        FDP_ADD(10, tp1);
        const unsigned long long tq1 = FDP_T();
        __syncthreads();  // the previous tile's LDS is dead (and the parameter block is visible)
        FDP_ADD(11, tq1);
        const unsigned long long tq2 = FDP_T();
        static_for<0, CIN>([&](auto ll) {
            constexpr int i = decltype(ll)::value;
            constexpr int DW = fd_lat_p(i) / 4, P = fd_lat_p(i);
            float* const dst = fd_smem + (i == 0 ? L.lat0 : L.rest + fd_lat_off_rel(i));
            const int oy = ay[i] - 4, ox = (ax[i] - 5) | 1;
            static_for<0, fd_s1_rounds(i)>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const int t = tid + k * kFdThreads;
                const int r = t / DW, j = t - r * DW;
                const int y = oy + r, x = ox + 4 * j;
                const int xs = fd_clamp(x, 0, max(gw[i] - 4, 0));
                const bool row_ok = y >= 0 && y < gh[i];
                const uint32_t w = ld_val[fd_s1_slot(i) + k];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sh = (x + e - xs) * 8;  // byte of the loaded dword that holds sample x + e (if it is in the grid)
                    const int b = static_cast<int>(static_cast<int8_t>(w >> (sh & 24)));
                    v[e] = (row_ok && x + e >= 0 && x + e < gw[i]) ? static_cast<float>(b) : 0.0f;
                }
                if (t < fd_s1_items(i)) *reinterpret_cast<f32x4*>(dst + r * P + 4 * j) = v;
            });
        });
        FDP_ADD(12, tq2);
        const unsigned long long tq3 = FDP_T();
        __syncthreads();
        FDP_ADD(13, tq3);
        FDP_ADD(1, tp1);
        const unsigned long long tp2 = FDP_T();

        // ---- S2 phase A: channel i of level i on footprint i, levels 1 .. L-1 (upsampling.py:486-498) ------------------------
        static_for<1, CIN>([&](auto ll) {
            constexpr int i = decltype(ll)::value;
            constexpr int PL = fd_lat_p(i), PD = fd_reg_w(i);
            const float* const lat = fd_smem + L.rest + fd_lat_off_rel(i);
            float* const dst = fd_smem + (i == 1 ? L.pc1 : L.rest + fd_pc_off_rel(CIN, i));
            const int oy = ay[i] - 4, ox = (ax[i] - 5) | 1;
            if constexpr (i == CIN - 1) {
                // the coarsest level enters the pyramid as it is; replicate outside the grid
                const float inv_fw = 1.0f / static_cast<float>(fw[i]);
                for (int j = (wave - fd_rot_a(CIN, i)) & 3; j < fd_wi_a(CIN, i); j += 4) {
                    const int e = j * 64 + lane;
                    int r = static_cast<int>((static_cast<float>(e) + 0.5f) * inv_fw), c = e - r * fw[i];
                    if (c < 0) { --r; c += fw[i]; }
                    if (c >= fw[i]) { ++r; c -= fw[i]; }
                    const int cy = fd_clamp(ay[i] + r, 0, gh[i] - 1), cx = fd_clamp(ax[i] + c, 0, gw[i] - 1);
                    if (e < fh[i] * fw[i]) dst[r * PD + c] = lat[(cy - oy) * PL + (cx - ox)];
                }
            } else {
                const int qy0 = ay[i] >> 1, qx0 = ax[i] >> 1;
                const int nqy = ((ay[i] + fh[i] - 1) >> 1) - qy0 + 1, nqx = ((ax[i] + fw[i] - 1) >> 1) - qx0 + 1, nq = nqy * nqx;
                const float inv_nqx = 1.0f / static_cast<float>(nqx);
                float wt[4];
                bool have_wt = false;
                for (int j = (wave - fd_rot_a(CIN, i)) & 3; j < fd_wi_a(CIN, i); j += 4) {
                    if (j * 64 >= nq) break;
                    if (!have_wt) { fd_preconv_weights(s_k2 + (i * 2 + 1) * 12, wpack, wt); have_wt = true; }
                    const int q_raw = j * 64 + lane, q = min(q_raw, nq - 1);  // surplus lanes recompute the last quad and store nothing
                    int qr = static_cast<int>((static_cast<float>(q) + 0.5f) * inv_nqx), qc = q - qr * nqx;
                    if (qc < 0) { --qr; qc += nqx; }
                    if (qc >= nqx) { ++qr; qc -= nqx; }
                    const int qy = qy0 + qr, qx = qx0 + qc;
                    const int cqy = bord[i] ? fd_clamp(qy, 0, (gh[i] - 1) >> 1) : qy, cqx = bord[i] ? fd_clamp(qx, 0, (gw[i] - 1) >> 1) : qx;
                    const f32x4 o4 = fd_preconv_quad<PL>(lat + (2 * cqy - 3 - oy) * PL + (2 * cqx - 3 - ox), wt);
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int y = 2 * qy + dy, x = 2 * qx + dx, r = y - ay[i], c = x - ax[i];
                            if (q_raw < nq && r >= 0 && r < fh[i] && c >= 0 && c < fw[i])
                                dst[r * PD + c] = bord[i] ? fd_pick(o4, fd_clamp(y, 0, gh[i] - 1), fd_clamp(x, 0, gw[i] - 1)) : o4[dy * 2 + dx];
                        }
                }
            }
        });
        FDP_ADD(14, tp2);
        // ---- S2 phase B: level i = L-2 .. 1: channels > i from level i + 1 through the x2 filter; wave-item = (channel, 64 quads).
        // Level i holds channel i in its own slot and channels > i in stack A (odd levels) or B (even levels).
        static_for_down<CIN - 2, 1>([&](auto ll) {
            constexpr int i = decltype(ll)::value;
            constexpr int PS = fd_reg_w(i + 1), PD = fd_reg_w(i), NCH = CIN - 1 - i, WQ = fd_wi_q(i);
            __syncthreads();
            float* const dst = fd_smem + ((i & 1) ? L.va : L.vb);
            const float* const src_stack = fd_smem + ((i & 1) ? L.vb : L.va);
            const float* const src_own = fd_smem + (i + 1 == 1 ? L.pc1 : L.rest + fd_pc_off_rel(CIN, i + 1));
            const int qy0 = ay[i] >> 1, qx0 = ax[i] >> 1;
            const int nqy = ((ay[i] + fh[i] - 1) >> 1) - qy0 + 1, nqx = ((ax[i] + fw[i] - 1) >> 1) - qx0 + 1, nq = nqy * nqx;
            const float inv_nqx = 1.0f / static_cast<float>(nqx);
            float wt[2];
            fd_tconv_weights(s_k2 + (i * 2) * 12, wpack, wt);
#pragma unroll 1
            for (int t = (wave + i) & 3; t < NCH * WQ; t += 4) {
                const int ch = t / WQ, j = t - ch * WQ;
                if (j * 64 >= nq) continue;
                const int q_raw = j * 64 + lane, q = min(q_raw, nq - 1);
                int qr = static_cast<int>((static_cast<float>(q) + 0.5f) * inv_nqx), qc = q - qr * nqx;
                if (qc < 0) { --qr; qc += nqx; }
                if (qc >= nqx) { ++qr; qc -= nqx; }
                const int qy = qy0 + qr, qx = qx0 + qc;
                const int cqy = bord[i] ? fd_clamp(qy, 0, (gh[i] - 1) >> 1) : qy, cqx = bord[i] ? fd_clamp(qx, 0, (gw[i] - 1) >> 1) : qx;
                // channel i + 1 + ch of level i + 1 (its own slot for ch == 0, else entry ch - 1 of the other stack) -> entry ch of this stack
                const float* sp = (ch == 0 ? src_own : src_stack + (ch - 1) * fd_pl(i + 1)) + (cqy - 2 - ay[i + 1]) * PS + (cqx - 2 - ax[i + 1]);
                const f32x4 o4 = fd_tconv_quad<PS>(sp, wt);
                float* dp = dst + ch * fd_pl(i);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int y = 2 * qy + dy, x = 2 * qx + dx, r = y - ay[i], c = x - ax[i];
                        if (q_raw < nq && r >= 0 && r < fh[i] && c >= 0 && c < fw[i])
                            dp[r * PD + c] = bord[i] ? fd_pick(o4, fd_clamp(y, 0, gh[i] - 1), fd_clamp(x, 0, gw[i] - 1)) : o4[dy * 2 + dx];
                    }
            }
        });
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        FDP_ADD(2, tp2);
        const unsigned long long tp3 = FDP_T();

        // ---- S3: level 0 in registers + the 1x1 layers -------------------------------------------------------------------
        // wave-pass = 32 x 2 quads (64 x 4 pixels); 8 passes per tile, wave w takes passes w and w + 4
        f32x4 stab[2][4][CT];  // stabiliser sums of the lane's pixels, kept for the epilogue
        float* const tile_a = fd_smem + L.tile_a;
        float* const tile_b = fd_smem + L.tile_b;
        const bool bord0 = bord[0];

        // + stabiliser, output transform (synthesis.py:286-294), stores of the lane's 2 x 2 pixels
        auto epilogue = [&](int pass, f32x4 (&y)[4][CT], int ey, int ex) {
            if (has_stab) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < C; ++j) y[s][j / 4][j % 4] = y[s][j / 4][j % 4] + stab[pass][s][j / 4][j % 4];
            }
            float wo[NWO];
#pragma unroll
            for (int v = 0; v < NWO; ++v) wo[v] = s_par[out_off + v * 64 + lane];
            f32x4 res[4][CT];
            static_for<0, CT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + outb_off + 4 * t);
                static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; res[s][t] = mstep<t>(wo, y[s][0][0], bias); });
                static_for<1, C>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; res[s][t] = mstep<i * CT + t>(wo, y[s][i / 4][i % 4], res[s][t]); });
                });
            });
            const int gy = ty0 + ey, gx = tx0 + ex;
            // interior of the tile and inside the image only
            if (ey < RM || ey >= kFdEH - RM || ex < RM || ex >= kFdEW - RM || gy < 0 || gx < 0 || gy >= H || gx >= W) return;
            const size_t plane = static_cast<size_t>(H) * W;
            const bool two_cols = gx + 1 < W, two_rows = gy + 1 < H;
            if (out_f32) {
                float __attribute__((address_space(1)))* const dst = (float __attribute__((address_space(1)))*)out_f32;
#pragma unroll
                for (int j = 0; j < C; ++j)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        if (dy == 1 && !two_rows) continue;
                        const size_t idx = j * plane + static_cast<size_t>(gy + dy) * W + gx;
                        dst[idx] = res[dy * 2][j / 4][j % 4];
                        if (two_cols) dst[idx + 1] = res[dy * 2 + 1][j / 4][j % 4];
                    }
            }
            if (write_planes == 2) {
                // yuv420 (decode.py:191-206, yuv.py:295): every sample onto the bit-depth grid FIRST, then U and V = the mean of
                // the lane's 2 x 2 quad as F.avg_pool2d forms it (sequential f32 sum in (dy, dx) order, / 4), clamp, grid again.
                // The quad is the lane's own (tile origins and ey / ex are even): nothing crosses lanes.  Same operations as
                // planes_kernel (ccd_float.hip), which served these frames through the f32 output before.
                const int ch = H >> 1, cw = W >> 1;
                if constexpr (C >= 3) {
                    if ((gy >> 1) < ch && (gx >> 1) < cw) {
#pragma unroll
                        for (int j = 1; j < 3; ++j) {
                            float sum = 0.0f;
#pragma unroll
                            for (int sq = 0; sq < 4; ++sq) sum += rintf(maxv * res[sq][j / 4][j % 4]) / maxv;
                            float a = sum / 4.0f;
                            a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
                            a = rintf(a * maxv) / maxv;
                            const unsigned q = static_cast<unsigned>(rintf(a * maxv));
                            const size_t cidx = static_cast<size_t>(gy >> 1) * cw + (gx >> 1);
                            if (bitdepth == 8) ((uint8_t __attribute__((address_space(1)))*)plane_ptr[j])[cidx] = static_cast<uint8_t>(q);
                            else ((uint16_t __attribute__((address_space(1)))*)plane_ptr[j])[cidx] = static_cast<uint16_t>(q);
                        }
                    }
                }
            }
            if (write_planes) {  // integer samples: rgb / yuv444 all three planes, yuv420 the luma plane
#pragma unroll
                for (int j = 0; j < (C < 3 ? C : 3); ++j)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        if ((dy == 1 && !two_rows) || (write_planes == 2 && j > 0)) continue;
                        const size_t idx = static_cast<size_t>(gy + dy) * W + gx;
                        const unsigned q0 = fd_quantise(res[dy * 2][j / 4][j % 4], maxv), q1 = fd_quantise(res[dy * 2 + 1][j / 4][j % 4], maxv);
                        if (bitdepth == 8) {
                            uint8_t __attribute__((address_space(1)))* const d8 = (uint8_t __attribute__((address_space(1)))*)plane_ptr[j];
                            d8[idx] = static_cast<uint8_t>(q0);
                            if (two_cols) d8[idx + 1] = static_cast<uint8_t>(q1);
                        } else {
                            uint16_t __attribute__((address_space(1)))* const d16 = (uint16_t __attribute__((address_space(1)))*)plane_ptr[j];
                            d16[idx] = static_cast<uint16_t>(q0);
                            if (two_cols) d16[idx + 1] = static_cast<uint16_t>(q1);
                        }
                    }
            }
        };
        // values of the lane's quad -> a 3x3-layer tile; a position outside the image takes the value of its clamped position
        auto store_tile = [&](float* tl, const f32x4 (&y)[4][CT], int ey, int ex) {
            const int gy = ty0 + ey, gx = tx0 + ex;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cy = fd_clamp(gy + (s >> 1), 0, H - 1), cx = fd_clamp(gx + (s & 1), 0, W - 1);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    float v = y[s][j / 4][j % 4];
                    if (bord0) {
                        const int i4 = ((cy & 1) << 1) | (cx & 1);
                        v = i4 == 0 ? y[0][j / 4][j % 4] : (i4 == 1 ? y[1][j / 4][j % 4] : (i4 == 2 ? y[2][j / 4][j % 4] : y[3][j / 4][j % 4]));
                    }
                    tl[j * kFdTileCh + (ey + (s >> 1) + 1) * kFdTP + ex + (s & 1) + 1] = v;
                }
            }
        };

        float wt_u0[2], wt_p0[4];  // A operands of the level-0 filters
        fd_tconv_weights(s_k2, wpack, wt_u0);
        fd_preconv_weights(s_k2 + 12, wpack, wt_p0);
        unsigned long long tp3b = 0, tp4 = 0;
        (void)tp3b; (void)tp4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int qrow = 2 * (wave + 4 * pass) + qyl;       // quad row / col inside the extended tile
            const int ey = 2 * qrow, ex = 2 * qxl;
            float x[CIN][4];                                   // dense values of the lane's 4 pixels
            {
                // the quad, clamped into the image in border tiles (replicate padding of the 3x3 layers: see store_tile)
                const int QY0 = (ty0 >> 1) + qrow, QX0 = (tx0 >> 1) + qxl;
                const int QY = bord0 ? fd_clamp(QY0, 0, (H - 1) >> 1) : QY0, QX = bord0 ? fd_clamp(QX0, 0, (W - 1) >> 1) : QX0;
                const int base1 = (QY - 2 - ay[1]) * fd_reg_w(1) + (QX - 2 - ax[1]);
                const float* const own1 = fd_smem + L.pc1 + base1;
                const float* const st1 = fd_smem + L.va + base1;
                // channels 1 .. CIN-1: level 1 through the x2 filter, two channels at a time
                static_for<0, (CIN - 1) / 2>([&](auto gg) {
                    constexpr int c0 = 1 + 2 * decltype(gg)::value, c1 = c0 + 1;
                    f32x4 r0, r1;
                    fd_tconv_quad2<fd_reg_w(1)>(c0 == 1 ? own1 : st1 + (c0 - 2) * fd_pl(1), st1 + (c1 - 2) * fd_pl(1), wt_u0, r0, r1);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) { x[c0][s4] = r0[s4]; x[c1][s4] = r1[s4]; }
                });
                if constexpr ((CIN - 1) % 2 == 1) {
                    constexpr int c0 = CIN - 1;
                    const f32x4 r0 = fd_tconv_quad<fd_reg_w(1)>(c0 == 1 ? own1 : st1 + (c0 - 2) * fd_pl(1), wt_u0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) x[c0][s4] = r0[s4];
                }
                // channel 0: the finest latent through the 7x7 filter
                const int oy = ay[0] - 4, ox = (ax[0] - 5) | 1;
                const f32x4 r0 = fd_preconv_quad<fd_lat_p(0)>(fd_smem + L.lat0 + (2 * QY - 3 - oy) * fd_lat_p(0) + (2 * QX - 3 - ox), wt_p0);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x[0][s4] = r0[s4];
            }
            FDP_ADD(3 + pass, pass == 0 ? tp3 : tp3b);
            tp4 = FDP_T();
            // ---- stabiliser on the raw inputs (synthesis.py:286-289), kept in registers until the epilogue
            if (has_stab) {
                float ws[NWS];
#pragma unroll
                for (int v = 0; v < NWS; ++v) ws[v] = s_par[stab_off + v * 64 + lane];
                static_for<0, CT>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + stabb_off + 4 * t);
                    static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; stab[pass][s][t] = mstep<t>(ws, x[0][s], bias); });
                    static_for<1, CIN>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; stab[pass][s][t] = mstep<c * CT + t>(ws, x[c][s], stab[pass][s][t]); });
                    });
                });
            }
            // ---- first and second 1x1 layers: hidden units in tiles of 4, never materialised beyond one tile
            f32x4 o[4][CT];
            static_for<0, CT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_par + b1_off + 4 * t);
#pragma unroll
                for (int s = 0; s < 4; ++s) o[s][t] = b1;
            });
#pragma unroll 1
            for (int n = 0; n < n_ht; ++n) {
                float wv[NWV];
#pragma unroll
                for (int v = 0; v < NWV; ++v) wv[v] = s_par[wq_off + (n * NWV + v) * 64 + lane];
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_par + b0_off + 4 * n);
                f32x4 d[4];
                static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; d[s] = mstep<0>(wv, x[0][s], b0); });
                static_for<1, CIN>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; d[s] = mstep<c>(wv, x[c][s], d[s]); });
                });
                if (relu0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[s][r] = fd_relu(d[s][r]);
                }
                static_for<0, 4>([&](auto rr) {
                    constexpr int r = decltype(rr)::value;
                    static_for<0, CT>([&](auto tt) {
                        constexpr int t = decltype(tt)::value;
                        static_for<0, 4>([&](auto ss) { constexpr int s = decltype(ss)::value; o[s][t] = mstep<CIN + t * 4 + r>(wv, d[s][r], o[s][t]); });
                    });
                });
            }
            if (relu1) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < CT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[s][t][r] = fd_relu(o[s][t][r]);
            }
            if (n_conv > 0) store_tile(tile_a, o, ey, ex);
            else epilogue(pass, o, ey, ex);
            FDP_ADD(5 + pass, tp4);
            tp3b = FDP_T();
        }
        const unsigned long long tp7 = FDP_T();

        // ---- S4: 3x3 layers on the LDS tiles (replicate padding was written by the producer); the last one runs the epilogue
        const float* cur = tile_a;
        float* nxt = tile_b;
        for (int l = 0; l < n_conv; ++l) {
            __syncthreads();
            const bool fin = l == n_conv - 1;
            float wc[NWC];
#pragma unroll
            for (int v = 0; v < NWC; ++v) wc[v] = s_par[p.conv_off[l] + v * 64 + lane];
            const int residual = p.conv_residual[l], relu = p.conv_relu[l], cb_off = p.convb_off[l];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int qrow = 2 * (wave + 4 * pass) + qyl;
                const int ey = 2 * qrow, ex = 2 * qxl;
                // the quad (clamped into the image in border tiles) and its 4 x 4 window, rows ey - 1 .. ey + 2 = tile rows ey .. ey + 3
                const int cey = bord0 ? fd_clamp(ty0 + ey, 0, (H - 1) & ~1) - ty0 : ey, cex = bord0 ? fd_clamp(tx0 + ex, 0, (W - 1) & ~1) - tx0 : ex;
                const float* const wb = cur + cey * kFdTP + cex;
                float win[C][4][4];
#pragma unroll
                for (int ci = 0; ci < C; ++ci)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const f32x2 t2 = *reinterpret_cast<const f32x2*>(wb + ci * kFdTileCh + a * kFdTP + 2 * b);
                            win[ci][a][2 * b] = t2[0]; win[ci][a][2 * b + 1] = t2[1];
                        }
                f32x4 y[4][CT];
                static_for<0, CT>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(s_par + cb_off + 4 * t);
#pragma unroll
                    for (int s = 0; s < 4; ++s) y[s][t] = bias;
                });
                static_for<0, 9 * C>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    constexpr int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
                    static_for<0, CT>([&](auto tt) {
                        constexpr int t = decltype(tt)::value;
                        static_for<0, 4>([&](auto ss) {
                            constexpr int s = decltype(ss)::value;
                            y[s][t] = mstep<k * CT + t>(wc, win[ci][(s >> 1) + ky][(s & 1) + kx], y[s][t]);
                        });
                    });
                });
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        float v = y[s][j / 4][j % 4];
                        if (residual) v = v + win[j][(s >> 1) + 1][(s & 1) + 1];
                        if (relu) v = fd_relu(v);
                        y[s][j / 4][j % 4] = v;
                    }
                if (!fin) store_tile(nxt, y, ey, ex);
                else epilogue(pass, y, ey, ex);
            }
            const float* tswap = cur; cur = nxt; nxt = const_cast<float*>(tswap);
        }
        FDP_ADD(7, tp7);
        FDP_ADD(8, tp1);
#ifdef CCD_FD_PROFILE
        if (tid == 0) atomicAdd(&fd_prof[9], 1ull);
#endif
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
int fused_dec_profile(unsigned long long* out16, int reset) {
#ifdef CCD_FD_PROFILE
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fd_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fd_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 1;
#else
    (void)out16; (void)reset;
    return 0;
#endif
}

bool fused_dec_supports(int c_in, int c) { return c_in >= 5 && c_in <= 9 && c >= 2 && c <= 5; }

size_t fused_dec_lds_bytes(int n_lv, int c, int n_conv, int n_params) {
    (void)n_conv;
    return static_cast<size_t>(fd_layout(n_lv, c).par + ((n_params + 3) & ~3)) * 4;
}

void fused_dec_param_shape(int c_in, int c, int* nwv, int* nws, int* nwc, int* nwo) {
    const int ct = (c + 3) / 4;
    *nwv = (c_in + 4 * ct + 15) / 16; *nws = (c_in * ct + 15) / 16; *nwc = (9 * c * ct + 15) / 16; *nwo = (c * ct + 15) / 16;
}

template <int CIN, int C>
static hipError_t launch_fd(const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<CIN, C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_fused_kernel<CIN, C>), dim3(n_work), dim3(kFdThreads), lds, stream, d_frames, d_work);
    return hipGetLastError();
}

template <int CIN>
static hipError_t launch_fd_c(int c, const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    switch (c) {
        case 2: return launch_fd<CIN, 2>(d_frames, d_work, n_work, lds, stream);
        case 3: return launch_fd<CIN, 3>(d_frames, d_work, n_work, lds, stream);
        case 4: return launch_fd<CIN, 4>(d_frames, d_work, n_work, lds, stream);
        case 5: return launch_fd<CIN, 5>(d_frames, d_work, n_work, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

// All frames of one launch share (c_in, c); `d_work` lists (frame, first tile, tile count) per workgroup.
hipError_t launch_fused_dec(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, size_t lds_bytes, hipStream_t stream) {
    if (n_work <= 0) return hipSuccess;
    const FdWork* w = static_cast<const FdWork*>(d_work);
    switch (c_in) {
        case 5: return launch_fd_c<5>(c, d_frames, w, n_work, lds_bytes, stream);
        case 6: return launch_fd_c<6>(c, d_frames, w, n_work, lds_bytes, stream);
        case 7: return launch_fd_c<7>(c, d_frames, w, n_work, lds_bytes, stream);
        case 8: return launch_fd_c<8>(c, d_frames, w, n_work, lds_bytes, stream);
        case 9: return launch_fd_c<9>(c, d_frames, w, n_work, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ccd
