// ccd_device.hpp - structures shared between the host API and the HIP kernels.
#pragma once

#include <cstdint>

#include "../../include/ccd.h"

namespace ccd {

constexpr int kMaxCtx = 40;          // spatial context template size (arm.py:501-509)
constexpr int kAcLo = -64;           // symbols live in [-64, 63] (constants.py:11)
constexpr int kAlphabet = 128;
constexpr int kRcPrecision = 24;     // constriction default range-coder precision
constexpr int kMuOffset = 16384;     // -MU_MIN_FIXED_POINT (constants.py:31)
constexpr int kScaleOffset = 1280;   // -LOG_SCALE_MIN_FIXED_POINT (constants.py:36)
constexpr int kNumMu = 32768, kNumScale = 2561;

// Per-slot description of the entropy stage (device memory, read-only for the kernel).
struct EntropyParams {
    const uint32_t* words;   // latent payload as little-endian u32 words
    uint32_t n_words;
    int32_t n_grids;
    int32_t grid_h[CCD_MAX_GRIDS], grid_w[CCD_MAX_GRIDS];
    int8_t* latent[CCD_MAX_GRIDS];     // decoded grids [h][w]
    int32_t ifce_in[CCD_MAX_GRIDS];    // number of IFCE input channels of the grid (0 = none)
    int32_t ifce_off[CCD_MAX_GRIDS];   // offset (in int64) of the grid's IFCE parameters in `ifce`
    int32_t level[CCD_MAX_GRIDS];      // number of size changes between grid 0 and grid g
    int32_t dim, n_spatial, n_ifce_out, n_layers, narrow, has_ifce;
    int32_t ctx_dy[kMaxCtx], ctx_dx[kMaxCtx];
    const int64_t* arm;      // per layer: w[in][out], b[out]; then ws[dim][2], bs[2]
    int32_t arm_len;         // length of `arm` in int64
    const int64_t* ifce;     // per grid with IFCE: w[in][out], b[out]
    int32_t* ifce_feat;      // scratch [n_ifce_out][fh][fw] (features at the previous grid's size)
    const float* scale_table;  // 2561 float32 scales
    const double* rcp_table;   // RN(1 / (double)scale): correctly rounded reciprocals for the f64 quotient
    int32_t* status;         // [0] error code, [1] words consumed, [2..3] symbols decoded (lo, hi)
    int32_t ring_rows;       // rows of the decoded-symbol ring in LDS: power of two >= widest grid / 10 + 6
    // Pipelined kernel, dynamic operand check (ccd_entropy_pipe.hip, "exactness"): features are kept as int16 planes in
    // `ifce_feat`; a feature with |f| >= 2^feat_bits is stored as the sentinel -32768 there and in full in `ifce_wide`
    // (int32 planes of the same shape); a pixel that meets a sentinel is redone in plain int64.  15 in production; tests
    // lower it (8..14) to force the redo path on ordinary streams.
    int32_t* ifce_wide;
    int32_t feat_bits;
    int32_t ifce_w32;        // every IFCE weight fits int32: the register-resident feature pass may be used
    int32_t mfma;            // > 0: the ARM's layers run on the matrix cores (limb-split int8, ccd_entropy_pipe.hip); the value
                             // is the number of bits a hidden activation may have before the task is redone in int64 (23)
};

// Upsampling level: stack_in [c_in][h_in][w_in] f32 (or the coarsest int8 grid) ->
// stack_out [c_in + 1][h_out][w_out]; channel 0 = pre-concat conv of the int8 grid `target`.
struct UpsampleLevel {
    const float* in_f32;     // nullptr when the input is the coarsest latent itself
    const int8_t* in_i8;
    const int8_t* target;    // [h_out][w_out]
    float* out;
    int32_t c_in, h_in, w_in, h_out, w_out;
    int32_t ups_k, pre_k;
    float ups_w[16], pre_w[16];
};

// Fused synthesis (ccd_synth_fused.hip): [N-1x1] [C-1x1] then up to 3 k x k layers on C channels,
// stabiliser and output transform.  All offsets index the float blob `params`.
struct SynthFused {
    const float* dense;      // [c_in][h][w]
    float* out;              // [c][h][w] synthesis output (f32) or null
    void* plane[3];          // integer planes (u8 / u16), used when write_planes
    const float* params;
    int32_t h, w, c_in, c, n_hidden;
    int32_t relu0, relu1;
    int32_t w0_off, b0_off;  // [n_hidden][CP], [n_hidden]   (CP = c_in rounded up to 4, zero padded)
    int32_t w1_off, b1_off;  // [c][n_hidden], [c]
    int32_t n_conv;
    int32_t conv_k[3], conv_residual[3], conv_relu[3], conv_w_off[3], conv_b_off[3];
    int32_t has_stab, stab_c_in, stab_w_off, stab_b_off;  // [c][CP] zero padded, [c]
    int32_t out_w_off, out_b_off;                         // [c][c], [c]
    int32_t halo;            // sum of the conv radii
    int32_t bitdepth, write_planes;
};

// Whole float path of a cool-chic in ONE kernel (ccd_fused.hip): int8 latent pyramid -> learned upsampling ->
// synthesis -> float and / or integer samples.  Nothing but the int8 grids is read from HBM.
constexpr int kFdMaxLevels = 12;     // latent levels (4K "auto" rule: 9)
constexpr int kFdMaxConv = 3;
// index of the kron product w[a] * w[b] (a, b = position folded onto the first half of the symmetric 1-D filter)
__host__ __device__ constexpr int k2_index(int a, int b) {
    return a <= b ? a * 4 - a * (a - 1) / 2 + (b - a) : b * 4 - b * (b - 1) / 2 + (a - b);
}
struct FusedDec {
    const int8_t* lat[kFdMaxLevels];  // latent (non-hyper) grids, finest first
    int32_t lh[kFdMaxLevels], lw[kFdMaxLevels];
    int32_t n_lv;
    // products of the symmetric 1-D filters, f32-rounded like the 2-D kron kernel the reference materialises
    // (upsampling.py:189-196, 312-325): k2u[i] = x2 transposed conv from level i + 1 to level i, k2p[i] = pre-concatenation
    // conv applied at level i.  10 distinct values each for k = 8 / k = 7 (k2_index).
    float k2u[kFdMaxLevels][10], k2p[kFdMaxLevels][10];
    // synthesis parameters in MFMA order (one "quad" = the 4 output-row weights of one multiply-add step), all offsets
    // in floats into `params`; the whole block [0, n_params) is staged in LDS
    const float* params;
    int32_t n_params;
    int32_t n_tiles_hidden;           // hidden units / 4 (rounded up)
    int32_t wq_off, b0_off;           // [n_tiles_hidden][NWV * 64], [n_tiles_hidden][4]
    int32_t b1_off;                   // [CT][4]
    int32_t stab_off, stabb_off;      // [NWS * 64], [CT][4]
    int32_t conv_off[kFdMaxConv], convb_off[kFdMaxConv];  // [NWC * 64], [CT][4]
    int32_t out_off, outb_off;        // [NWO * 64], [CT][4]
    int32_t h, w;                     // size of the finest latent level = size of the synthesis output
    int32_t c;                        // output channels
    int32_t relu0, relu1, n_conv, conv_residual[kFdMaxConv], conv_relu[kFdMaxConv], has_stab;
    int32_t margin;                   // halo of the tile, even, >= n_conv
    int32_t tiles_x, tiles_y;
    float* out;                       // [c][h][w] f32 or null
    void* plane[3];
    int32_t bitdepth, write_planes;
    // PRE = true instantiations: levels >= 1 evaluated once per frame by the batch's pyramid steps; channels 1 .. n_lv - 1 at the
    // resolution of level 1, f32 [n_lv - 1][lh[1]][lw[1]] (channel 1 = level 1's own latent through the 7x7 filter first)
    const float* l1;
    // common randomness (NZ = n_lv instantiations): the noise planes at full resolution, f32 [n_lv][h][w]
    const float* noise;
};

}  // namespace ccd
