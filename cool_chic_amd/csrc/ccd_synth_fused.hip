// ccd_synth_fused.hip - the synthesis of a whole frame in ONE kernel: dense latent planes in, RGB/YUV
// float planes and integer planes out, nothing else touches HBM (algorithmic traffic 4 L + 4 C (+ C)
// bytes per pixel).
//
// Covers the architecture family every decoder preset of the reference uses (cfg/dec/*/*.cfg):
//     [N-1-linear-{relu,none}], [C-1-linear-{relu,none}], then 0..3 layers C-k-{residual,linear}-{relu,none}
// plus the linear stabiliser and the output transform (component/core/synthesis.py:272-294).  Anything
// else runs through the generic per-layer kernels of ccd_float.hip.
//
// Structure: a 256-thread workgroup owns a 64x32 "extended" tile (interior + halo R = sum of the conv
// radii).  Phase 1 evaluates the two 1x1 layers per pixel without ever materialising the N hidden
// channels: hidden unit h is reduced into the C outputs as soon as it is computed, which is exactly
// the oracle's accumulation order (ci ascending).  Weights are wave-uniform: they are read with
// scalar loads and feed the FMAs as SGPR operands, 4 pixels per thread share each load.  The C-channel
// result goes to an LDS tile; the k x k layers ping-pong between two LDS tiles (replicate padding =
// clamping the image coordinate before it is turned into a tile coordinate); the epilogue adds the
// stabiliser, applies the output transform and writes float and integer samples.
//
// Numerics: every multiply-add is an explicit __fmaf_rn in the oracle's order -> bit-identical.
#include <hip/hip_runtime.h>

#include "ccd_device.hpp"

namespace ccd {

constexpr int kSfThreads = 256;
constexpr int kSfEW = 64, kSfEH = 32;              // extended tile
constexpr int kSfPos = kSfEW * kSfEH;              // 2048 positions, 8 per thread
constexpr int kSfPerThread = kSfPos / kSfThreads;  // 8
constexpr int kSfGroup = 4;                        // pixels evaluated together in phase 1

__device__ __forceinline__ int sf_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float sf_round_to_grid(float x, float maxv) { return rintf(maxv * x) / maxv; }
__device__ __forceinline__ unsigned sf_quantise(float x, float maxv) {
    float q = sf_round_to_grid(x, maxv);
    q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    q = rintf(q * maxv) / maxv;
    return static_cast<unsigned>(rintf(q * maxv));
}

// CP = input channels padded to a multiple of 4 (extra inputs are 0 with 0 weights: fma(0, 0, acc) == acc),
// C = channels after the second 1x1 layer (= channels of the conv layers and of the output).
template <int CP, int C>
__global__ __launch_bounds__(kSfThreads) void syn_fused_kernel(const SynthFused* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float sf_lds[];
    const SynthFused& p = frames[blockIdx.z];  // one frame per z-slice: many frames share one launch
    {
        const int tiles_x = (p.w + (kSfEW - 2 * p.halo) - 1) / (kSfEW - 2 * p.halo);
        const int tiles_y = (p.h + (kSfEH - 2 * p.halo) - 1) / (kSfEH - 2 * p.halo);
        if (static_cast<int>(blockIdx.x) >= tiles_x || static_cast<int>(blockIdx.y) >= tiles_y) return;  // grid is sized for the largest frame
    }
    float* bufA = sf_lds;                // [C][kSfEH][kSfEW]
    float* bufB = sf_lds + C * kSfPos;
    const int tid = threadIdx.x;
    const int R = p.halo;
    const int iw = kSfEW - 2 * R, ih = kSfEH - 2 * R;  // interior size
    const int tx0 = blockIdx.x * iw - R, ty0 = blockIdx.y * ih - R;  // image coordinate of extended (0, 0)
    const int H = p.h, W = p.w;
    const size_t plane = static_cast<size_t>(H) * W;
    // explicit global address space: through the descriptor these are generic pointers, whose FLAT loads / stores also
    // count on lgkmcnt - every wait for an LDS tile would wait for the HBM traffic as well
    typedef const float __attribute__((address_space(1)))* gcf_t;
    const gcf_t dense = (gcf_t)p.dense;
    const gcf_t prm = (gcf_t)p.params;

    // ---- phase 1: the two 1x1 layers, per pixel, evaluated at clamped image coordinates -------------
    {
        const gcf_t w0 = prm + p.w0_off;  // [N][CP]
        const gcf_t b0 = prm + p.b0_off;
        const gcf_t w1 = prm + p.w1_off;  // [C][N]
        const gcf_t b1 = prm + p.b1_off;
        const int N = p.n_hidden;
#pragma unroll 1
        for (int g = 0; g < kSfPerThread / kSfGroup; ++g) {
            float x[kSfGroup][CP];
            float o[kSfGroup][C];
#pragma unroll
            for (int u = 0; u < kSfGroup; ++u) {
                const int pos = tid + kSfThreads * (g * kSfGroup + u);
                const int ey = pos / kSfEW, ex = pos % kSfEW;
                const int gy = sf_clamp(ty0 + ey, 0, H - 1), gx = sf_clamp(tx0 + ex, 0, W - 1);
                const gcf_t src = dense + static_cast<size_t>(gy) * W + gx;
#pragma unroll
                for (int c = 0; c < CP; ++c) x[u][c] = c < p.c_in ? src[c * plane] : 0.0f;
#pragma unroll
                for (int j = 0; j < C; ++j) o[u][j] = b1[j];
            }
            for (int h = 0; h < N; ++h) {
                float a[kSfGroup];
                const float bh = b0[h];
#pragma unroll
                for (int u = 0; u < kSfGroup; ++u) a[u] = bh;
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    const float w = w0[h * CP + c];
#pragma unroll
                    for (int u = 0; u < kSfGroup; ++u) a[u] = __fmaf_rn(w, x[u][c], a[u]);
                }
                if (p.relu0) {
#pragma unroll
                    for (int u = 0; u < kSfGroup; ++u) a[u] = a[u] <= 0.0f ? 0.0f : a[u];  // NaN stays NaN like torch.relu
                }
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const float w = w1[j * N + h];
#pragma unroll
                    for (int u = 0; u < kSfGroup; ++u) o[u][j] = __fmaf_rn(w, a[u], o[u][j]);
                }
            }
#pragma unroll
            for (int u = 0; u < kSfGroup; ++u) {
                const int pos = tid + kSfThreads * (g * kSfGroup + u);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    float v = o[u][j];
                    if (p.relu1) v = v <= 0.0f ? 0.0f : v;
                    bufA[j * kSfPos + pos] = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: k x k layers on the LDS tiles (replicate padding at the image border) ----------------
    float* cur = bufA;
    float* nxt = bufB;
    for (int l = 0; l < p.n_conv; ++l) {
        const int k = p.conv_k[l], pad = (k - 1) / 2;
        const gcf_t wl = prm + p.conv_w_off[l];  // [C][C][k][k]
        const gcf_t bl = prm + p.conv_b_off[l];
        const int residual = p.conv_residual[l], relu = p.conv_relu[l];
#pragma unroll 1
        for (int m = 0; m < kSfPerThread; ++m) {
            const int pos = tid + kSfThreads * m;
            const int ey = pos / kSfEW, ex = pos % kSfEW;
            const int gy = ty0 + ey, gx = tx0 + ex;
            float acc[C];
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j] = bl[j];
            // outside the image nothing is ever read back (readers clamp first), and close to the tile edge the
            // taps would leave the tile: those positions belong to the halo consumed by this layer.
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W && ey >= pad && ey < kSfEH - pad && ex >= pad && ex < kSfEW - pad;
            if (inside && k == 3) {
                // 3x3: clamp the three rows / columns once, then 9 LDS reads per input channel
                int sy[3], sx[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    sy[d] = (sf_clamp(gy + d - 1, 0, H - 1) - ty0) * kSfEW;
                    sx[d] = sf_clamp(gx + d - 1, 0, W - 1) - tx0;
                }
#pragma unroll
                for (int ci = 0; ci < C; ++ci) {
                    float v[9];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = cur[ci * kSfPos + sy[ky] + sx[kx]];
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int j = 0; j < C; ++j) acc[j] = __fmaf_rn(wl[(j * C + ci) * 9 + t], v[t], acc[j]);
                }
            } else if (inside) {
                for (int ci = 0; ci < C; ++ci) {
                    for (int ky = 0; ky < k; ++ky) {
                        const int sy = sf_clamp(gy + ky - pad, 0, H - 1) - ty0;
                        for (int kx = 0; kx < k; ++kx) {
                            const int sx = sf_clamp(gx + kx - pad, 0, W - 1) - tx0;
                            const float v = cur[ci * kSfPos + sy * kSfEW + sx];
#pragma unroll
                            for (int j = 0; j < C; ++j) acc[j] = __fmaf_rn(wl[((j * C + ci) * k + ky) * k + kx], v, acc[j]);
                        }
                    }
                }
            }
            if (inside) {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    float v = acc[j];
                    if (residual) v = v + cur[j * kSfPos + pos];
                    if (relu) v = v <= 0.0f ? 0.0f : v;
                    nxt[j * kSfPos + pos] = v;
                }
            }
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
    }

    // ---- epilogue: + stabiliser, output transform, stores ------------------------------------------------
    const gcf_t ws = prm + p.stab_w_off;  // [C][c_in]
    const gcf_t bs = prm + p.stab_b_off;
    const gcf_t wo = prm + p.out_w_off;   // [C][C]
    const gcf_t bo = prm + p.out_b_off;
    const float maxv = static_cast<float>((1 << p.bitdepth) - 1);
#pragma unroll 1
    for (int m = 0; m < kSfPerThread; ++m) {
        const int pos = tid + kSfThreads * m;
        const int ey = pos / kSfEW, ex = pos % kSfEW;
        const int gy = ty0 + ey, gx = tx0 + ex;
        if (ey < R || ey >= kSfEH - R || ex < R || ex >= kSfEW - R || gy >= H || gx >= W) continue;
        float y[C];
#pragma unroll
        for (int j = 0; j < C; ++j) y[j] = cur[j * kSfPos + pos];
        if (p.has_stab) {
            const gcf_t src = dense + static_cast<size_t>(gy) * W + gx;
            float xs[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) xs[c] = c < p.stab_c_in ? src[c * plane] : 0.0f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                float s = bs[j];
#pragma unroll
                for (int c = 0; c < CP; ++c) s = __fmaf_rn(ws[j * CP + c], xs[c], s);  // padded weights are 0
                y[j] = y[j] + s;
            }
        }
        float out[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float s = bo[j];
#pragma unroll
            for (int i = 0; i < C; ++i) s = __fmaf_rn(wo[j * C + i], y[i], s);
            out[j] = s;
        }
        const size_t idx = static_cast<size_t>(gy) * W + gx;
        if (p.out) {
#pragma unroll
            for (int j = 0; j < C; ++j) ((float __attribute__((address_space(1)))*)p.out)[j * plane + idx] = out[j];
        }
        if (p.write_planes) {  // rgb / yuv444 integer samples (decode.py:191-206); yuv420 goes through planes_kernel
#pragma unroll
            for (int j = 0; j < (C < 3 ? C : 3); ++j) {
                const unsigned q = sf_quantise(out[j], maxv);
                if (p.bitdepth == 8) ((uint8_t __attribute__((address_space(1)))*)p.plane[j])[idx] = static_cast<uint8_t>(q);
                else ((uint16_t __attribute__((address_space(1)))*)p.plane[j])[idx] = static_cast<uint16_t>(q);
            }
        }
    }
}

template <int CP, int C>
static hipError_t launch_one(const SynthFused* d_frames, int n_frames, dim3 grid, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(syn_fused_kernel<CP, C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    grid.z = n_frames;
    hipLaunchKernelGGL((syn_fused_kernel<CP, C>), grid, dim3(kSfThreads), lds, stream, d_frames);
    return hipGetLastError();
}

template <int CP>
static hipError_t launch_cp(int c, const SynthFused* d_frames, int n_frames, dim3 grid, size_t lds, hipStream_t stream) {
    switch (c) {
        case 2: return launch_one<CP, 2>(d_frames, n_frames, grid, lds, stream);
        case 3: return launch_one<CP, 3>(d_frames, n_frames, grid, lds, stream);
        case 4: return launch_one<CP, 4>(d_frames, n_frames, grid, lds, stream);
        case 5: return launch_one<CP, 5>(d_frames, n_frames, grid, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

bool syn_fused_supports(int c_in, int c, int halo) {
    return c_in >= 1 && c_in <= 16 && c >= 2 && c <= 5 && halo >= 0 && 2 * halo < kSfEH - 8;
}

void syn_fused_tiles(int h, int w, int halo, int* tiles_x, int* tiles_y) {
    *tiles_x = (w + (kSfEW - 2 * halo) - 1) / (kSfEW - 2 * halo);
    *tiles_y = (h + (kSfEH - 2 * halo) - 1) / (kSfEH - 2 * halo);
}

// All frames of one launch share (CP, C); `d_frames` is a device array, the grid covers the largest frame.
hipError_t launch_syn_fused(const SynthFused* d_frames, int n_frames, int c_in, int c, int max_tiles_x, int max_tiles_y,
                            hipStream_t stream) {
    if (n_frames <= 0) return hipSuccess;
    dim3 grid(max_tiles_x, max_tiles_y, 1);
    const size_t lds = static_cast<size_t>(2) * c * kSfPos * sizeof(float);
    switch ((c_in + 3) / 4) {
        case 1: return launch_cp<4>(c, d_frames, n_frames, grid, lds, stream);
        case 2: return launch_cp<8>(c, d_frames, n_frames, grid, lds, stream);
        case 3: return launch_cp<12>(c, d_frames, n_frames, grid, lds, stream);
        case 4: return launch_cp<16>(c, d_frames, n_frames, grid, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ccd
