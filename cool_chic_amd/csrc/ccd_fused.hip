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
// the oracle's order (oracle/cc_oracle.c sections 8-9).  The synthesis chains run on the matrix cores:
// v_mfma_f32_4x4x1_16b_f32 computes D[i] = fma(A[i], B, C[i]) per lane with ONE rounding per product and accumulates in
// issue order (tools/ubench/mfma_probe.hip: 0 of 51 200 words differ from the __fmaf_rn chain), so a chain of such
// instructions IS the fmaf chain, four output channels at a time.  The upsampling chains are explicit __fmaf_rn on the
// vector ALU.  The file is compiled with -ffp-contract=off.
//
// Structure.  A 256-thread workgroup owns a 64 x 32 "extended" tile of the finest level (interior + an even halo margin
// >= the number of 3x3 layers) and walks the pyramid coarse -> fine inside LDS:
//   S1  the int8 latents of every level's footprint -> f32 tiles in LDS (zero outside the grid: the pre-concatenation
//       conv pads with zeros);
//   S2  level i = L-2 .. 1: every channel c >= i on the level's footprint from level i + 1 (x2 transposed conv, replicate
//       padding = clamped coordinates) and the level's own latent (7x7 conv + residual), ping-pong between two LDS stacks.
//       A level needs rows [q - 2, q + 2] of the level below for its quad rows q, so footprints shrink by half and stop
//       at ~10 x 10: recomputing the coarse levels per tile costs a few % of the tile;
//   S3  level 0: a lane owns a 2 x 2 quad of pixels (= 4 "sets" of 64 pixels per wave, lane = pixel in each set) and
//       computes its L dense values per pixel in registers; they go straight into the matrix cores as B operands:
//       first 1x1 layer (hidden units in tiles of 4 = the 4 rows of the MFMA), ReLU, second 1x1 layer and the
//       stabiliser, all in registers.  The A operands (weights) are one VGPR per 16 multiply-add steps: CBSZ / ABID
//       broadcast one of the 16 blocks' A rows to all blocks, so the 4 weights of step q sit in lanes 4 (q % 16) .. + 3;
//   S4  the 3x3 layers ping-pong between two LDS tiles; each tap is one MFMA step whose B operand is a register of the
//       lane's 4 x 4 window; the last layer continues into the stabiliser add, the output transform (MFMA again) and
//       the stores.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ccd_device.hpp"

namespace ccd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFdThreads = 256;
constexpr int kFdEW = 64, kFdEH = 32;  // extended tile of the finest level
constexpr int kFdTilePx = kFdEW * kFdEH;

// Largest footprint of level i for a 64 x 32 tile whose origin is even (rows, cols); see the header comment.
__host__ __device__ constexpr int fd_reg_h(int i) { return i == 0 ? 32 : i == 1 ? 20 : i == 2 ? 15 : i == 3 ? 12 : i == 4 ? 11 : 10; }
__host__ __device__ constexpr int fd_reg_w(int i) { return i == 0 ? 64 : i == 1 ? 36 : i == 2 ? 23 : i == 3 ? 16 : i == 4 ? 13 : i == 5 ? 11 : 10; }
__host__ __device__ constexpr int fd_lat_elems(int i) { return ((fd_reg_h(i) + 6) * (fd_reg_w(i) + 6) + 3) & ~3; }

__host__ __device__ constexpr int fd_lat_prefix(int i) { int n = 0; for (int j = 1; j < i; ++j) n += fd_lat_elems(j); return n; }  // levels 1 .. i - 1

struct FdLayout {  // offsets in 4-byte words into the workgroup's dynamic LDS (scalars only: no dynamically indexed arrays)
    int geom, k2, params, lat0, lat_rest, va, vb, pc, tile_a, tile_b, total;
};
// Channel i of level i (the level's own pre-concatenation conv, or the latent itself at the coarsest level) has its own
// slot per level: all of them are produced up front in one phase.
__host__ __device__ constexpr int fd_pc_prefix(int i) { int n = 0; for (int j = 1; j < i; ++j) n += fd_reg_h(j) * fd_reg_w(j); return n; }

__host__ __device__ inline FdLayout fd_layout(int n_lv, int c, int n_conv, int n_params) {
    FdLayout L;
    int o = 0;
    L.geom = o; o += 2 * 8 * kFdMaxLevels;
    L.k2 = o; o += 2 * 10 * kFdMaxLevels;
    L.params = o; o += (n_params + 3) & ~3;
    L.lat0 = o; o += fd_lat_elems(0);
    L.va = o; o += (n_lv > 2 ? n_lv - 2 : 0) * fd_reg_h(1) * fd_reg_w(1);   // level 1 (3, 5, ..): channels 2 .. L-1
    L.pc = o; o += fd_pc_prefix(n_lv);
    const int mid = o;
    L.lat_rest = o; o += fd_lat_prefix(n_lv);
    L.vb = o; o += (n_lv > 3 ? n_lv - 3 : 0) * fd_reg_h(2) * fd_reg_w(2);   // level 2 (4, 6, ..): channels 3 .. L-1
    // conv tiles alias the pyramid: A is written at the end of S3 (the coarse levels are dead), B only in S4
    const int tile = c * kFdTilePx;
    L.tile_a = mid; L.tile_b = L.lat0;
    if (n_conv >= 2 && mid - L.lat0 < tile) L.tile_a = L.lat0 + tile;
    if (n_conv >= 1 && L.tile_a + tile > o) o = L.tile_a + tile;
    L.total = o;
    return L;
}
__host__ __device__ inline int fd_pc_off(const FdLayout& L, int i) {  // LDS offset of channel i of level i (i >= 1)
    int n = 0;
    for (int j = 1; j < kFdMaxLevels; ++j) n += j < i ? fd_reg_h(j) * fd_reg_w(j) : 0;
    return L.pc + n;
}
__host__ __device__ inline int fd_lat_off(const FdLayout& L, int i) {  // LDS offset of level i's latent tile
    if (i == 0) return L.lat0;
    int n = 0;
    for (int j = 1; j < kFdMaxLevels; ++j) n += j < i ? fd_lat_elems(j) : 0;
    return L.lat_rest + n;
}

extern __shared__ __attribute__((aligned(16))) float fd_smem[];

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
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


// The upsampling filters run on the matrix cores too: a lane owns a 2 x 2 output quad, the 4 rows of the MFMA are the 4
// outputs (row = 2 dy + dx), one step per sample of the lane's source window.  A sample that an output does not use has
// weight 0 for that row: fma(v, 0, acc) == acc bit for bit (v is finite, acc is never -0), so every output still sees
// exactly its own taps, in its own order.
//
// x2 transposed conv (k = 8, replicate pad 4, crop 11): 5 x 5 window v[a][b] = source (qy - 2 + a, qx - 2 + b).  Output row
// 2 qy uses ky = 1, 3, 5, 7 on window rows 3, 2, 1, 0, output row 2 qy + 1 uses ky = 0, 2, 4, 6 on rows 4, 3, 2, 1 (same
// for columns): taps in ky, kx ascending order = window rows and columns DESCENDING.  Step s <-> (a, b) = (4 - s / 5, 4 - s % 5).
__device__ __forceinline__ void fd_tconv_weights(const float* k2 /*10 kron products, LDS*/, int lane, float (&wt)[2]) {
    const int ph = lane & 3, dy = ph >> 1, dx = ph & 1;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int st = 16 * v + (lane >> 2);
        const int a = 4 - st / 5, b = 4 - st % 5;
        const int ky = dy == 0 ? 2 * (3 - a) + 1 : 2 * (4 - a), kx = dx == 0 ? 2 * (3 - b) + 1 : 2 * (4 - b);
        const bool ok = st < 25 && ky >= 0 && ky < 8 && kx >= 0 && kx < 8;
        const int fy = ky < 4 ? ky : 7 - ky, fx = kx < 4 ? kx : 7 - kx;  // symmetric filter (a b c d d c b a)
        wt[v] = ok ? k2[k2_index(fy & 3, fx & 3)] : 0.0f;
    }
}
__device__ __forceinline__ f32x4 fd_tconv_quad(const float (&v)[5][5], const float (&wt)[2]) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<0, 25>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        acc = mstep<st>(wt, v[4 - st / 5][4 - st % 5], acc);
    });
    return acc;
}
// Pre-concatenation 7x7 conv (zero padding, + residual): 8 x 8 window v[a][b] = latent (2 qy - 3 + a, 2 qx - 3 + b), zeros
// outside the grid (a zero sample contributes nothing, which is the oracle's skipping of those taps).  Output (dy, dx)
// uses tap (ky, kx) = (a - dy, b - dx); step s <-> (a, b) = (s / 8, s % 8), ascending.
__device__ __forceinline__ void fd_preconv_weights(const float* k2 /*10 kron products, LDS*/, int lane, float (&wt)[4]) {
    const int ph = lane & 3, dy = ph >> 1, dx = ph & 1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int st = 16 * v + (lane >> 2);
        const int ky = st / 8 - dy, kx = st % 8 - dx;
        const bool ok = ky >= 0 && ky < 7 && kx >= 0 && kx < 7;
        const int fy = ky < 4 ? ky : 6 - ky, fx = kx < 4 ? kx : 6 - kx;  // symmetric filter (a b c d c b a)
        wt[v] = ok ? k2[k2_index(fy & 3, fx & 3)] : 0.0f;
    }
}
__device__ __forceinline__ f32x4 fd_preconv_quad(const float (&v)[8][8], const float (&wt)[4]) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<0, 64>([&](auto ss) {
        constexpr int st = decltype(ss)::value;
        acc = mstep<st>(wt, v[st / 8][st % 8], acc);
    });
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) acc[ph] = acc[ph] + v[(ph >> 1) + 3][(ph & 1) + 3];
    return acc;
}

// S1 maps the threads of the workgroup onto a level's latent tile with a power-of-two row pitch: level 0 (tile <= 38 x 70)
// 2 rows x 128 columns per round, level 1 (<= 26 x 42) 4 x 64, levels 2 .. 5 8 x 32, deeper levels 16 x 16.
__host__ __device__ constexpr int fd_s1_shift(int i) { return i == 0 ? 7 : i == 1 ? 6 : i <= 5 ? 5 : 4; }
__host__ __device__ constexpr int fd_s1_rounds(int i) { return ((fd_reg_h(i) + 6) * (1 << fd_s1_shift(i)) + 255) / 256; }
__host__ __device__ constexpr int fd_s1_slot(int i) { int n = 0; for (int j = 0; j < i; ++j) n += fd_s1_rounds(j); return n; }

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

// CIN = latent levels = input channels of the synthesis, C = its output channels (both fix register arrays and the
// immediate operands of the MFMA steps).
template <int CIN, int C>
__global__ __launch_bounds__(kFdThreads, 2) void decode_fused_kernel(const FusedDec* __restrict__ frames, const FdWork* __restrict__ work) {
    constexpr int CT = (C + 3) / 4;                  // output-channel tiles of 4
    constexpr int NWV = (CIN + 4 * CT + 15) / 16;    // weight registers per hidden tile: CIN first-layer steps + 4 CT second-layer steps
    constexpr int NWS = (CIN * CT + 15) / 16;        // stabiliser
    constexpr int NWC = (9 * C * CT + 15) / 16;      // one 3x3 layer
    constexpr int NWO = (C * CT + 15) / 16;          // output transform
    typedef const float __attribute__((address_space(1)))* gcf_t;
    typedef const int8_t __attribute__((address_space(1)))* gci8_t;

    const FdWork wk = work[blockIdx.x];
    const FusedDec& p = frames[wk.frame];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_lv = p.n_lv;
    const int H = p.h, W = p.w;
    const FdLayout L = fd_layout(n_lv, C, p.n_conv, p.n_params);
    int* const s_geom2 = reinterpret_cast<int*>(fd_smem + L.geom);
    float* const s_k2 = fd_smem + L.k2;       // [level][x2 filter | pre-concatenation filter][10]
    float* const s_par = fd_smem + L.params;

    // ---- parameters -> LDS once per workgroup (MFMA order, see FusedDec) --------------------------------------------
    const unsigned long long tp0 = FDP_T();
    {
        const gcf_t src = (gcf_t)p.params;
        for (int i = tid * 4; i < p.n_params; i += kFdThreads * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(src + i);
            *reinterpret_cast<f32x4*>(s_par + i) = v;
        }
    }
    for (int i = tid; i < n_lv * 20; i += kFdThreads) {
        const int lvl = i / 20, j = i - lvl * 20;
        s_k2[i] = j < 10 ? p.k2u[lvl][j] : p.k2p[lvl][j - 10];
    }
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

    for (int tile = wk.tile_first; tile < wk.tile_first + wk.tile_count; ++tile) {
        const int by = tile / p.tiles_x, bx = tile - by * p.tiles_x;
        const int tx0 = bx * iw - RM, ty0 = by * ih - RM;  // image coordinate of extended (0, 0); both even
        const unsigned long long tp1 = FDP_T();
        int* const s_geom = s_geom2 + (tile & 1) * (8 * kFdMaxLevels);  // double-buffered: written before the barrier below
        // ---- footprints + S1 loads.  Level 0 = the tile clipped to the image; level i + 1 = rows / cols [q0 - 2, q1 + 2]
        // clipped to the grid (grid i = ceil(grid 0 / 2^i)).  The footprints are wave-uniform scalar arithmetic; the latent
        // bytes of every level's footprint (+ 3 on every side for the 7x7 conv, zero outside the grid) are requested
        // back-to-back from clamped addresses (a conditional load makes the compiler wait at every join) and stored as f32
        // after ONE wait.
        int lv_th[kFdMaxLevels], lv_tw[kFdMaxLevels];      // tile size per level (static indices only)
        int wi_off[kFdMaxLevels + 1];                       // first phase-A wave-item of level i (levels 1 .. n_lv - 1)
        int ld_val[fd_s1_slot(kFdMaxLevels)];
        unsigned long long ld_ok = 0;  // bit s: load slot s lies inside its grid (else the tile holds the zero padding)
        static_assert(fd_s1_slot(kFdMaxLevels) <= 64, "one validity bit per load slot");
        {
            int y0 = max(ty0, 0), y1 = min(ty0 + kFdEH - 1, H - 1), x0 = max(tx0, 0), x1 = min(tx0 + kFdEW - 1, W - 1);
            int n_wi = 0;
            static_for<0, kFdMaxLevels>([&](auto ll) {
                constexpr int lvl = decltype(ll)::value;
                constexpr int SH = fd_s1_shift(lvl);
                wi_off[lvl] = n_wi;
                lv_th[lvl] = 0; lv_tw[lvl] = 1;
                if (lvl < n_lv) {
                    const int gh = (H + (1 << lvl) - 1) >> lvl, gw = (W + (1 << lvl) - 1) >> lvl;
                    const int rh = y1 - y0 + 1, rw = x1 - x0 + 1, th = rh + 6, tw = rw + 6;
                    const int qy0 = y0 >> 1, qx0 = x0 >> 1, nqy = (y1 >> 1) - qy0 + 1, nqx = (x1 >> 1) - qx0 + 1;
                    if (tid == 0) {
                        int* g = s_geom + 8 * lvl;
                        g[0] = y0; g[1] = rh; g[2] = x0; g[3] = rw; g[4] = qy0; g[5] = nqx; g[6] = qx0; g[7] = nqy * nqx;
                    }
                    // phase A of S2 in wave-items of 64: pre-concatenation conv quads of levels 1 .. n_lv - 2, samples of the coarsest
                    if (lvl >= 1) n_wi += ((lvl == n_lv - 1 ? rh * rw : nqy * nqx) + 63) >> 6;
                    lv_th[lvl] = th; lv_tw[lvl] = tw;
                    const gci8_t src = (gci8_t)p.lat[lvl];
                    const int c = tid & ((1 << SH) - 1), rb = tid >> SH;
                    const int x = x0 - 3 + c;
                    const bool col_ok = c < tw && x >= 0 && x < gw;
                    const int xc = fd_clamp(x, 0, gw - 1);
                    static_for<0, fd_s1_rounds(lvl)>([&](auto kk) {
                        constexpr int k = decltype(kk)::value;
                        const int r = k * (kFdThreads >> SH) + rb, y = y0 - 3 + r;
                        if (col_ok && r < th && y >= 0 && y < gh) ld_ok |= 1ull << (fd_s1_slot(lvl) + k);
                        ld_val[fd_s1_slot(lvl) + k] = src[static_cast<size_t>(fd_clamp(y, 0, gh - 1)) * gw + xc];
                    });
                    if (lvl + 1 < n_lv) {
                        const int hn = (H + (2 << lvl) - 1) >> (lvl + 1), wn = (W + (2 << lvl) - 1) >> (lvl + 1);
                        y0 = max((y0 >> 1) - 2, 0); y1 = min((y1 >> 1) + 2, hn - 1);
                        x0 = max((x0 >> 1) - 2, 0); x1 = min((x1 >> 1) + 2, wn - 1);
                    }
                }
            });
            wi_off[kFdMaxLevels] = n_wi;
        }
        FDP_ADD(10, tp1);
        const unsigned long long tq1 = FDP_T();
        __syncthreads();  // the previous tile's LDS is dead (and the parameter block / this tile's footprints are visible)
        FDP_ADD(11, tq1);
        const unsigned long long tq2 = FDP_T();
        static_for<0, kFdMaxLevels>([&](auto ll) {
            constexpr int lvl = decltype(ll)::value;
            constexpr int SH = fd_s1_shift(lvl);
            if (lvl < n_lv) {
                float* const dst = fd_smem + fd_lat_off(L, lvl);
                const int c = tid & ((1 << SH) - 1), rb = tid >> SH;
                static_for<0, fd_s1_rounds(lvl)>([&](auto kk) {
                    constexpr int k = decltype(kk)::value;
                    const int r = k * (kFdThreads >> SH) + rb;
                    if (c < lv_tw[lvl] && r < lv_th[lvl])
                        dst[r * lv_tw[lvl] + c] = ((ld_ok >> (fd_s1_slot(lvl) + k)) & 1ull) ? static_cast<float>(ld_val[fd_s1_slot(lvl) + k]) : 0.0f;
                });
            }
        });
        FDP_ADD(12, tq2);
        const unsigned long long tq3 = FDP_T();
        __syncthreads();
        FDP_ADD(13, tq3);
        FDP_ADD(1, tp1);
        const unsigned long long tp2 = FDP_T();
        // ---- S2: the coarse levels.  Level i holds channels i .. L-1 on footprint i: channel i in its own slot, channels > i in
        // stack A (odd levels) or B (even levels).  A lane owns a 2 x 2 quad; a wave-item = 64 quads of ONE level (the filter is
        // the A operand of the MFMA, shared by the wave); surplus lanes recompute the last quad and store nothing.
        // Phase A: channel i of every level i >= 1 depends on the level's own latent only: pre-concatenation conv (7x7 + residual)
        // for i <= L-2, the latent itself at the coarsest level (upsampling.py:486-498).
        {
            const int total = wi_off[kFdMaxLevels];
#pragma unroll 1
            for (int wi = wave; wi < total; wi += kFdThreads / 64) {
                int lvl = 1;
                static_for<2, kFdMaxLevels>([&](auto ll) { constexpr int l2 = decltype(ll)::value; lvl += (l2 < n_lv && wi >= wi_off[l2]) ? 1 : 0; });
                int wi0 = 0;
                static_for<1, kFdMaxLevels>([&](auto ll) { constexpr int l2 = decltype(ll)::value; wi0 = l2 == lvl ? wi_off[l2] : wi0; });
                lvl = __builtin_amdgcn_readfirstlane(lvl);
                const int* g = s_geom + 8 * lvl;
                const int ry0 = g[0], rh = g[1], rx0 = g[2], rw = g[3], qy0 = g[4], nqx = g[5], qx0 = g[6], nq = g[7];
                const float* const lat = fd_smem + fd_lat_off(L, lvl);
                float* const dst = fd_smem + fd_pc_off(L, lvl);
                const int ltw = rw + 6;
                const int q_raw = (wi - wi0) * 64 + lane;
                if (lvl == n_lv - 1) {
                    if (q_raw < rh * rw) {
                        int r = static_cast<int>((static_cast<float>(q_raw) + 0.5f) / static_cast<float>(rw)), c = q_raw - r * rw;
                        if (c < 0) { --r; c += rw; }
                        if (c >= rw) { ++r; c -= rw; }
                        dst[q_raw] = lat[(r + 3) * ltw + c + 3];
                    }
                    continue;
                }
                const int q = min(q_raw, nq - 1);
                int qr = static_cast<int>((static_cast<float>(q) + 0.5f) / static_cast<float>(nqx)), qc = q - qr * nqx;
                if (qc < 0) { --qr; qc += nqx; }
                if (qc >= nqx) { ++qr; qc -= nqx; }
                const int qy = qy0 + qr, qx = qx0 + qc;
                float wt[4];
                fd_preconv_weights(s_k2 + (lvl * 2 + 1) * 10, lane, wt);
                float v[8][8];
                const int wy = 2 * qy - ry0, wx = 2 * qx - rx0;  // window origin (2 qy - 3, 2 qx - 3) in tile coordinates
                // a quad's first / last row or column may lie outside the footprint: the clamped reads only feed outputs
                // that are not stored
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const int ra = fd_clamp(wy + a, 0, rh + 5) * ltw;
#pragma unroll
                    for (int b = 0; b < 8; ++b) v[a][b] = lat[ra + fd_clamp(wx + b, 0, rw + 5)];
                }
                const f32x4 o4 = fd_preconv_quad(v, wt);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int r = 2 * qy + dy - ry0, cidx = 2 * qx + dx - rx0;
                        if (q_raw < nq && r >= 0 && r < rh && cidx >= 0 && cidx < rw) dst[r * rw + cidx] = o4[dy * 2 + dx];
                    }
            }
        }
        FDP_ADD(14, tp2);
        // Phase B: level i = L-2 .. 1: channels c > i from level i + 1 (x2 transposed conv at clamped coordinates);
        // wave-item = (channel, 64 quads)
        for (int i = n_lv - 2; i >= 1; --i) {
            __syncthreads();
            const int* g = s_geom + 8 * i;
            const int ry0 = g[0], rh = g[1], rx0 = g[2], rw = g[3], qy0 = g[4], nqx = g[5], qx0 = g[6], nq = g[7];
            const int sy0 = g[8], sh = g[9], sx0 = g[10], sw = g[11];
            float* const dst = fd_smem + ((i & 1) ? L.va : L.vb);
            const float* const src = fd_smem + ((i & 1) ? L.vb : L.va);
            const float* const src_pc = fd_smem + fd_pc_off(L, i + 1);
            const int hs = (H + (2 << i) - 1) >> (i + 1), ws = (W + (2 << i) - 1) >> (i + 1);
            float wt[2];
            fd_tconv_weights(s_k2 + (i * 2) * 10, lane, wt);
            const int n_ch = n_lv - 1 - i, wpc = (nq + 63) >> 6;  // wave-items per channel
            const int plane_d = rh * rw, plane_s = sh * sw;
            const float inv_nqx = 1.0f / static_cast<float>(nqx);
#pragma unroll 1
            for (int wi = wave; wi < wpc * n_ch; wi += kFdThreads / 64) {
                const int ch = wi / wpc, q_raw = (wi - ch * wpc) * 64 + lane;
                const int q = min(q_raw, nq - 1);
                int qr = static_cast<int>((static_cast<float>(q) + 0.5f) * inv_nqx), qc = q - qr * nqx;
                if (qc < 0) { --qr; qc += nqx; }
                if (qc >= nqx) { ++qr; qc -= nqx; }
                const int qy = qy0 + qr, qx = qx0 + qc;
                float v[5][5];
                // channel i + 1 + ch of level i + 1 (its own slot for ch == 0, else entry ch - 1 of the other stack) -> entry ch of this stack
                const float* sp = ch == 0 ? src_pc : src + (ch - 1) * plane_s;
                int ro[5], co[5];
#pragma unroll
                for (int d = 0; d < 5; ++d) {
                    ro[d] = fd_clamp(fd_clamp(qy - 2 + d, 0, hs - 1) - sy0, 0, sh - 1) * sw;
                    co[d] = fd_clamp(fd_clamp(qx - 2 + d, 0, ws - 1) - sx0, 0, sw - 1);
                }
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b < 5; ++b) v[a][b] = sp[ro[a] + co[b]];
                const f32x4 o4 = fd_tconv_quad(v, wt);
                float* dp = dst + ch * plane_d;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const int r = 2 * qy + dy - ry0, cidx = 2 * qx + dx - rx0;
                        if (q_raw < nq && r >= 0 && r < rh && cidx >= 0 && cidx < rw) dp[r * rw + cidx] = o4[dy * 2 + dx];
                    }
            }
        }
        __syncthreads();
        FDP_ADD(2, tp2);
        const unsigned long long tp3 = FDP_T();

        // ---- S3: level 0 in registers + the 1x1 layers on the matrix cores -------------------------------------------
        // wave-pass = 32 x 2 quads (64 x 4 pixels); 8 passes per tile, wave w takes passes w and w + 4
        f32x4 stab[2][4][CT];  // stabiliser sums of the lane's pixels, kept for the epilogue
        const int qxl = lane & 31, qyl = lane >> 5;
        const int ry0_0 = s_geom[0], rh_0 = s_geom[1], rx0_0 = s_geom[2], rw_0 = s_geom[3];
        const int lat0_w = rw_0 + 6;
        const int sy1 = s_geom[8], sh1 = s_geom[9], sx1 = s_geom[10], sw1 = s_geom[11];
        float* const tile_a = fd_smem + L.tile_a;
        float* const tile_b = fd_smem + L.tile_b;

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
            if (ey < RM || ey >= kFdEH - RM || ex < RM || ex >= kFdEW - RM || gy >= H || gx >= W) return;
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
            if (write_planes) {  // rgb / yuv444 integer samples; yuv420 goes through planes_kernel
#pragma unroll
                for (int j = 0; j < (C < 3 ? C : 3); ++j)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        if (dy == 1 && !two_rows) continue;
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

        float wt_u0[2], wt_p0[4];  // A operands of the level-0 filters
        fd_tconv_weights(s_k2, lane, wt_u0);
        fd_preconv_weights(s_k2 + 10, lane, wt_p0);
        unsigned long long tp3b = 0, tp4 = 0;
        (void)tp3b; (void)tp4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int qrow = 2 * (wave + 4 * pass) + qyl;       // quad row / col inside the extended tile
            const int ey = 2 * qrow, ex = 2 * qxl;
            float x[CIN][4];                                   // dense values of the lane's 4 pixels
            {
                // image quad coordinates (clamped in the halo outside the image: nothing reads those results back)
                const int QY = fd_clamp((ty0 >> 1) + qrow, 0, (H - 1) >> 1), QX = fd_clamp((tx0 >> 1) + qxl, 0, (W - 1) >> 1);
                const float* const lat0 = fd_smem + L.lat0;
                if constexpr (CIN > 1) {
                    const int hs = (H + 1) >> 1, ws = (W + 1) >> 1;
                    int ro[5], co[5];
#pragma unroll
                    for (int d = 0; d < 5; ++d) {
                        ro[d] = fd_clamp(fd_clamp(QY - 2 + d, 0, hs - 1) - sy1, 0, sh1 - 1) * sw1;
                        co[d] = fd_clamp(fd_clamp(QX - 2 + d, 0, ws - 1) - sx1, 0, sw1 - 1);
                    }
                    const float* const v1 = fd_smem + L.va;
                    const float* const pc1 = fd_smem + L.pc;
                    static_for<1, CIN>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        float v[5][5];
                        const float* sp = c == 1 ? pc1 : v1 + (c - 2) * (sh1 * sw1);
#pragma unroll
                        for (int a = 0; a < 5; ++a)
#pragma unroll
                            for (int b = 0; b < 5; ++b) v[a][b] = sp[ro[a] + co[b]];
                        const f32x4 o4 = fd_tconv_quad(v, wt_u0);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) x[c][s4] = o4[s4];
                    });
                    float v[8][8];
                    const int wy = 2 * QY - ry0_0, wx = 2 * QX - rx0_0;  // window origin (2 QY - 3, 2 QX - 3) in tile coordinates
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const int ra = fd_clamp(wy + a, 0, rh_0 + 5) * lat0_w;
#pragma unroll
                        for (int b = 0; b < 8; ++b) v[a][b] = lat0[ra + fd_clamp(wx + b, 0, rw_0 + 5)];
                    }
                    const f32x4 o4 = fd_preconv_quad(v, wt_p0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) x[0][s4] = o4[s4];
                } else {
                    // a single level: dense = float(latent) (coolchic.py:175-177 with nothing to upsample)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        x[0][s] = lat0[fd_clamp(2 * QY + (s >> 1) - ry0_0 + 3, 0, rh_0 + 5) * lat0_w + fd_clamp(2 * QX + (s & 1) - rx0_0 + 3, 0, rw_0 + 5)];
                }
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
            if (n_conv > 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < C; ++j) tile_a[j * kFdTilePx + (ey + (s >> 1)) * kFdEW + ex + (s & 1)] = o[s][j / 4][j % 4];
            } else {
                epilogue(pass, o, ey, ex);
            }
            FDP_ADD(5 + pass, tp4);
            tp3b = FDP_T();
        }
        const unsigned long long tp7 = FDP_T();

        // ---- S4: 3x3 layers on the LDS tiles (replicate padding = clamped image coordinates); the last one runs the epilogue
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
                const int gy = ty0 + ey, gx = tx0 + ex;
                // 4 x 4 window of the quad at clamped image coordinates, as tile coordinates (halo positions whose window
                // leaves the tile read a clamped copy: their results are never consumed)
                int ro[4], co[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    ro[d] = fd_clamp(fd_clamp(gy - 1 + d, 0, H - 1) - ty0, 0, kFdEH - 1) * kFdEW;
                    co[d] = fd_clamp(fd_clamp(gx - 1 + d, 0, W - 1) - tx0, 0, kFdEW - 1);
                }
                float win[C][4][4];
#pragma unroll
                for (int ci = 0; ci < C; ++ci)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) win[ci][a][b] = cur[ci * kFdTilePx + ro[a] + co[b]];
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
                if (!fin) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < C; ++j) nxt[j * kFdTilePx + (ey + (s >> 1)) * kFdEW + ex + (s & 1)] = y[s][j / 4][j % 4];
                } else {
                    epilogue(pass, y, ey, ex);
                }
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

size_t fused_dec_lds_bytes(int n_lv, int c, int n_conv, int n_params) { return static_cast<size_t>(fd_layout(n_lv, c, n_conv, n_params).total) * 4; }

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
