// ccd_float.hip - the float stages of the decoder: learned latent-pyramid upsampling, the synthesis
// conv stack and the integer output planes.
//
// Reference behaviour (paths relative to /root/reference/coolchic):
//   component/core/upsampling.py:463-500,287-330,158-203   Upsampling.forward (training-mode 2-D kernels)
//   component/core/synthesis.py:61-76,272-294               Synthesis.forward
//   bitstream/component/coolchic.py:187-192                 final resize + crop
//   bitstream/decode.py:191-206, io/format/{png,yuv}.py     rounding / 4:2:0 / clamping
//
// Numerics contract: every multiply-add below is an explicit __fmaf_rn in the SAME order as the CPU
// oracle (oracle/cc_oracle.c sections 8-9); the file is compiled with -ffp-contract=off so nothing
// else fuses.  Results are bit-identical to the oracle.
//
// This file holds the generic (any architecture the header can describe) kernels.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ccd_device.hpp"

namespace ccd {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// -------------------------------------------------------------------------------------------------
// Upsampling: one launch per pyramid level.  blockIdx.z = output channel: 0 is the pre-concat conv of
// the level's own latent, c >= 1 is the x2 transposed conv of input channel c-1.
// -------------------------------------------------------------------------------------------------
// Each thread produces a 2x2 output quad.  For the x2 transposed conv the four outputs of a quad read one
// shared (k/2+1)^2 window of the input (5x5 for k = 8) and the k x k kron products are formed once per
// thread and reused by every channel; accumulation order per output is unchanged (ky ascending, kx ascending).
// Global address space stated explicitly: through the level descriptor the pointers are generic and would be accessed with
// FLAT instructions (slower address path, and they count on lgkmcnt as well as vmcnt).
typedef const float __attribute__((address_space(1)))* gcf_t;
typedef const int8_t __attribute__((address_space(1)))* gci8_t;
typedef float __attribute__((address_space(1)))* gf_t;
// fast paths: ups_k == 8 (tconv8_quad) and pre_k == 7 (preconv7_quad); other kernel sizes use the generic body

__device__ __forceinline__ void upsample_generic_one(const UpsampleLevel& L, int ch, int ox, int oy) {
    float acc = 0.0f;
    if (ch == 0) {
        const int k = L.pre_k, pad = k / 2;
        const gci8_t t = (gci8_t)L.target;
        for (int ky = 0; ky < k; ++ky) {
            const int sy = oy + ky - pad;
            if (sy < 0 || sy >= L.h_out) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int sx = ox + kx - pad;
                if (sx < 0 || sx >= L.w_out) continue;
                const float k2 = L.pre_w[ky] * L.pre_w[kx];
                acc = __fmaf_rn(static_cast<float>(t[sy * L.w_out + sx]), k2, acc);
            }
        }
        acc = acc + static_cast<float>(t[oy * L.w_out + ox]);
    } else {
        const int k = L.ups_k, p0 = k / 2, crop = 2 * p0 - 1 + k / 2;
        const int h = L.h_in, w = L.w_in;
        const int py = oy + crop, px = ox + crop;
        const gcf_t inf = L.in_f32 ? (gcf_t)L.in_f32 + static_cast<size_t>(ch - 1) * h * w : (gcf_t) nullptr;
        for (int ky = py & 1; ky < k; ky += 2) {
            const int iy = (py - ky) / 2;
            if (py - ky < 0 || iy >= h + 2 * p0) continue;
            const int sy = clampi(iy - p0, 0, h - 1);
            for (int kx = px & 1; kx < k; kx += 2) {
                const int ix = (px - kx) / 2;
                if (px - kx < 0 || ix >= w + 2 * p0) continue;
                const int sx = clampi(ix - p0, 0, w - 1);
                const float k2 = L.ups_w[ky] * L.ups_w[kx];
                const float v = inf ? inf[sy * w + sx] : static_cast<float>(((gci8_t)L.in_i8)[sy * w + sx]);
                acc = __fmaf_rn(v, k2, acc);
            }
        }
    }
    ((gf_t)L.out)[(static_cast<size_t>(ch) * L.h_out + oy) * L.w_out + ox] = acc;
}

// x2 transposed conv, k == 8 (the value every preset uses): quad (qy, qx) -> outputs (2qy + dy, 2qx + dx).
// With crop 11 and pad 4: output row 2qy   (py odd ) uses ky = 1,3,5,7 on source rows qy+1, qy, qy-1, qy-2
//                         output row 2qy+1 (py even) uses ky = 0,2,4,6 on source rows qy+2, qy+1, qy, qy-1
__device__ __forceinline__ void tconv8_quad(const UpsampleLevel& L, int qx, int qy, int ch_first, int ch_last) {
    const int h = L.h_in, w = L.w_in;
    float wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = L.ups_w[i];
    int sy[5], sx[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        sy[d] = clampi(qy - 2 + d, 0, h - 1) * w;
        sx[d] = clampi(qx - 2 + d, 0, w - 1);
    }
    const int oy0 = 2 * qy, ox0 = 2 * qx;
    for (int ch = ch_first; ch <= ch_last; ++ch) {
        float v[5][5];
        if (L.in_f32) {
            const gcf_t src = (gcf_t)L.in_f32 + static_cast<size_t>(ch - 1) * h * w;
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) v[a][b] = src[sy[a] + sx[b]];
        } else {
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) v[a][b] = static_cast<float>(((gci8_t)L.in_i8)[sy[a] + sx[b]]);
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float acc = 0.0f;
#pragma unroll
                for (int ty = 0; ty < 4; ++ty) {
                    const int ky = (1 - dy) + 2 * ty;          // ascending ky
                    const int ra = (dy == 0 ? 3 : 4) - ty;     // row index in v (source row descends as ky ascends)
#pragma unroll
                    for (int tx = 0; tx < 4; ++tx) {
                        const int kx = (1 - dx) + 2 * tx;
                        const int cb = (dx == 0 ? 3 : 4) - tx;
                        acc = __fmaf_rn(v[ra][cb], wv[ky] * wv[kx], acc);
                    }
                }
                const int oy = oy0 + dy, ox = ox0 + dx;
                if (oy < L.h_out && ox < L.w_out) ((gf_t)L.out)[(static_cast<size_t>(ch) * L.h_out + oy) * L.w_out + ox] = acc;
            }
        }
    }
}

// Pre-concatenation 7x7 conv (zero padding, kron kernel, residual) for a 2x2 output quad: one 8x8 window of
// the int8 latent feeds the four outputs.  Out-of-range taps are loaded as 0: fma(0, k2, acc) == acc bit for bit
// (acc is never -0), which equals the oracle's skipping of those taps.
__device__ __forceinline__ void preconv7_quad(const UpsampleLevel& L, int qx, int qy) {
    const int h = L.h_out, w = L.w_out;
    const gci8_t t = (gci8_t)L.target;
    float wv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) wv[i] = L.pre_w[i];
    const int y0 = 2 * qy - 3, x0 = 2 * qx - 3;
    float v[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int yy = y0 + a;
        const bool row_ok = yy >= 0 && yy < h;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int xx = x0 + b;
            v[a][b] = (row_ok && xx >= 0 && xx < w) ? static_cast<float>(t[yy * w + xx]) : 0.0f;
        }
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            float acc = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) acc = __fmaf_rn(v[dy + ky][dx + kx], wv[ky] * wv[kx], acc);
            acc = acc + v[dy + 3][dx + 3];
            const int oy = 2 * qy + dy, ox = 2 * qx + dx;
            if (oy < h && ox < w) ((gf_t)L.out)[static_cast<size_t>(oy) * w + ox] = acc;
        }
    }
}

// One launch per pyramid step for ALL frames of a batch: z enumerates (frame, channel group) pairs through
// `zmap` (high 16 bits: index into `levels`, low 16 bits: channel group); the x/y grid covers the largest frame.
// Channel group 0 = pre-concat conv of the level's latent; group g >= 1 = transposed conv of input channels
// 2g-2 .. 2g-1 (two channels per thread share the kron products).
__global__ __launch_bounds__(256) void upsample_step_kernel(const UpsampleLevel* __restrict__ levels, const uint32_t* __restrict__ zmap) {
    const uint32_t z = zmap[blockIdx.z];
    const UpsampleLevel& L = levels[z >> 16];
    const int grp = static_cast<int>(z & 0xffffu);
    // a block covers 64 x 16 outputs = 32 x 8 quads
    if (static_cast<int>(blockIdx.x) * 64 >= L.w_out || static_cast<int>(blockIdx.y) * 16 >= L.h_out) return;
    const int qx = blockIdx.x * 32 + (threadIdx.x & 31), qy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (2 * qx >= L.w_out || 2 * qy >= L.h_out) return;
    if (grp == 0 && L.pre_k == 7) { preconv7_quad(L, qx, qy); return; }
    if (grp == 0) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                if (2 * qy + dy < L.h_out && 2 * qx + dx < L.w_out) upsample_generic_one(L, 0, 2 * qx + dx, 2 * qy + dy);
        return;
    }
    const int ch_first = 2 * grp - 1, ch_last = min(2 * grp, static_cast<int>(L.c_in));
    if (L.ups_k == 8) { tconv8_quad(L, qx, qy, ch_first, ch_last); return; }
    for (int ch = ch_first; ch <= ch_last; ++ch)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx)
                if (2 * qy + dy < L.h_out && 2 * qx + dx < L.w_out) upsample_generic_one(L, ch, 2 * qx + dx, 2 * qy + dy);
}

hipError_t launch_upsample_step(const UpsampleLevel* d_levels, const uint32_t* d_zmap, int n_z, int max_w, int max_h, hipStream_t stream) {
    if (n_z <= 0) return hipSuccess;
    dim3 grid((max_w + 63) / 64, (max_h + 15) / 16, n_z);
    hipLaunchKernelGGL(upsample_step_kernel, grid, dim3(256), 0, stream, d_levels, d_zmap);
    return hipGetLastError();
}

// Coarsest-and-only level (latent_resolution[0] == latent_resolution[1]): dense = float(latent).
__global__ void i8_to_f32_kernel(const int8_t* in, float* out, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<float>(in[i]);
}

hipError_t launch_i8_to_f32(const int8_t* in, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(i8_to_f32_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Generic synthesis layer: out[co] = act( bias[co] + sum_{ci,ky,kx} w * in_eff[ci](replicate pad) (+ in_eff[co]) )
// with in_eff = in (+ in2 when given: the "main + stabiliser" sum feeding the output transform).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void syn_layer_kernel(const float* __restrict__ in, const float* __restrict__ in2,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ out, int c_in, int c_out, int k, int residual,
                                                        int relu, int h, int w) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int pad = (k - 1) / 2;
    const size_t plane = static_cast<size_t>(h) * w;
    for (int co = 0; co < c_out; ++co) {
        float acc = bias[co];
        for (int ci = 0; ci < c_in; ++ci) {
            for (int ky = 0; ky < k; ++ky) {
                const int sy = clampi(y + ky - pad, 0, h - 1);
                for (int kx = 0; kx < k; ++kx) {
                    const int sx = clampi(x + kx - pad, 0, w - 1);
                    float v = in[ci * plane + static_cast<size_t>(sy) * w + sx];
                    if (in2) v = v + in2[ci * plane + static_cast<size_t>(sy) * w + sx];
                    acc = __fmaf_rn(wt[((static_cast<size_t>(co) * c_in + ci) * k + ky) * k + kx], v, acc);
                }
            }
        }
        if (residual) {
            float v = in[co * plane + static_cast<size_t>(y) * w + x];
            if (in2) v = v + in2[co * plane + static_cast<size_t>(y) * w + x];
            acc = acc + v;
        }
        if (relu) acc = acc <= 0.0f ? 0.0f : acc;  // NaN stays NaN like torch.relu (and in the oracle)
        out[co * plane + static_cast<size_t>(y) * w + x] = acc;
    }
}

hipError_t launch_syn_layer(const float* in, const float* in2, const float* wt, const float* bias, float* out, int c_in,
                            int c_out, int k, int residual, int relu, int h, int w, hipStream_t stream) {
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(syn_layer_kernel, grid, dim3(256), 0, stream, in, in2, wt, bias, out, c_in, c_out, k, residual,
                       relu, h, w);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Final resize (coolchic.py:187-192): nearest only (motion cool-chics); size-preserving modes are
// exact identities and are skipped by the host.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
    if (in_size == out_size) return dst;
    if (out_size == 2 * in_size) return dst >> 1;
    const float scale = static_cast<float>(in_size) / static_cast<float>(out_size);
    const int s = static_cast<int>(floorf(static_cast<float>(dst) * scale));
    return s < in_size - 1 ? s : in_size - 1;
}

__global__ void resize_nearest_kernel(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t n = static_cast<size_t>(c) * h_out * w_out;
    if (i >= n) return;
    const int x = static_cast<int>(i % w_out);
    const int y = static_cast<int>((i / w_out) % h_out);
    const int ch = static_cast<int>(i / (static_cast<size_t>(w_out) * h_out));
    out[i] = in[(static_cast<size_t>(ch) * h_in + nearest_src(y, h_in, h_out)) * w_in + nearest_src(x, w_in, w_out)];
}

hipError_t launch_resize_nearest(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out,
                                 hipStream_t stream) {
    const size_t n = static_cast<size_t>(c) * h_out * w_out;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, in, out,
                       c, h_in, w_in, h_out, w_out);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// F.interpolate(mode = bilinear | bicubic, align_corners=False): the final resize when the finest latent
// is coarser than the picture (coolchic.py:187-189; scale = in / out) and the x2 steps of
// fixed_upsampling(mode="bicubic") for the common-randomness planes (upsampling.py:556-595; scale 0.5).
// Same float sequence as the oracle (section 9b): plain ops for coordinates and coefficients,
// t_j = w_x0 v_0, fmaf(v_i, w_xi, t_j); out = w_y0 t_0, fmaf(t_j, w_yj, out).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int interp_taps(int dst, int in_size, float scale, int cubic, int idx[4], float wt[4]) {
    float src = scale * (static_cast<float>(dst) + 0.5f) - 0.5f;
    if (!cubic && src < 0.0f) src = 0.0f;
    int i0 = static_cast<int>(floorf(src));
    if (i0 > in_size - 1) i0 = in_size - 1;
    float t = src - static_cast<float>(i0);
    t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    if (!cubic) {
        idx[0] = i0; idx[1] = i0 + (i0 < in_size - 1 ? 1 : 0);
        wt[0] = 1.0f - t; wt[1] = t;
        return 2;
    }
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = (1.0f - t) + 1.0f;
    wt[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    wt[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    wt[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    wt[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int v = i0 - 1 + j; idx[j] = v < 0 ? 0 : (v > in_size - 1 ? in_size - 1 : v); }
    return 4;
}

__global__ void resize_interp_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int h_in, int w_in,
                                     int h_out, int w_out, int cubic, float scale_y, float scale_x) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w_out || y >= h_out) return;
    int iy[4], ix[4];
    float wy[4], wx[4];
    const int ny = interp_taps(y, h_in, scale_y, cubic, iy, wy);
    const int nx = interp_taps(x, w_in, scale_x, cubic, ix, wx);
    for (int ch = 0; ch < c; ++ch) {
        const float* plane = in + static_cast<size_t>(ch) * h_in * w_in;
        float acc = 0.0f;
        for (int j = 0; j < ny; ++j) {
            const float* row = plane + static_cast<size_t>(iy[j]) * w_in;
            float t = row[ix[0]] * wx[0];
            for (int i = 1; i < nx; ++i) t = __fmaf_rn(row[ix[i]], wx[i], t);
            acc = j == 0 ? t * wy[0] : __fmaf_rn(t, wy[j], acc);
        }
        out[(static_cast<size_t>(ch) * h_out + y) * w_out + x] = acc;
    }
}

hipError_t launch_resize_interp(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out, int cubic,
                                float scale_y, float scale_x, hipStream_t stream) {
    dim3 grid((w_out + 63) / 64, (h_out + 3) / 4);
    hipLaunchKernelGGL(resize_interp_kernel, grid, dim3(256), 0, stream, in, out, c, h_in, w_in, h_out, w_out, cubic,
                       scale_y, scale_x);
    return hipGetLastError();
}

// coolchic.py:187-189 for any final_upsampling_type (0 nearest, 1 bilinear, 2 bicubic)
hipError_t launch_final_resize(const float* in, float* out, int c, int h_in, int w_in, int h_out, int w_out, int mode,
                               hipStream_t stream) {
    if (mode == 0) return launch_resize_nearest(in, out, c, h_in, w_in, h_out, w_out, stream);
    return launch_resize_interp(in, out, c, h_in, w_in, h_out, w_out, mode == 2, static_cast<float>(h_in) / static_cast<float>(h_out),
                                static_cast<float>(w_in) / static_cast<float>(w_out), stream);
}

// -------------------------------------------------------------------------------------------------
// Common randomness (component/core/noise.py:17-54): Park-Miller LCG + Box-Muller, evaluated in f64
// exactly like the oracle (section 9c: fixed fma sequences for log and cos), one thread per sample with a
// jump-ahead of the generator: sample i uses draws 2i+1 and 2i+2, seed_k = a^k seed_0 mod m.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cr_sin_core(double r) {
    const double r2 = r * r;
    double p = -7.6471637318198164759e-13;
    p = fma(p, r2, 1.6059043836821614599e-10);
    p = fma(p, r2, -2.5052108385441718775e-08);
    p = fma(p, r2, 2.7557319223985890653e-06);
    p = fma(p, r2, -1.9841269841269841270e-04);
    p = fma(p, r2, 8.3333333333333333333e-03);
    p = fma(p, r2, -1.6666666666666666667e-01);
    return fma(p * r2, r, r);
}
__device__ __forceinline__ double cr_cos_core(double r) {
    const double r2 = r * r;
    double p = 4.7794773323873852974e-14;
    p = fma(p, r2, -1.1470745597729724714e-11);
    p = fma(p, r2, 2.0876756987868098979e-09);
    p = fma(p, r2, -2.7557319223985890653e-07);
    p = fma(p, r2, 2.4801587301587301587e-05);
    p = fma(p, r2, -1.3888888888888888889e-03);
    p = fma(p, r2, 4.1666666666666666667e-02);
    p = fma(p, r2, -0.5);
    return fma(p, r2, 1.0);
}
__device__ __forceinline__ double cr_log(double x) {
    unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(x));
    int e = static_cast<int>((bits >> 52) & 0x7ff) - 1022;
    bits = (bits & 0x000fffffffffffffULL) | 0x3fe0000000000000ULL;
    double f = __longlong_as_double(static_cast<long long>(bits));
    if (f < 0.70710678118654752440) { f = f * 2.0; e -= 1; }
    const double s = __ddiv_rn(f - 1.0, f + 1.0), s2 = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0); p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0); p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0);
    const double r = fma(p * s2, 2.0 * s, 2.0 * s);
    const double ed = static_cast<double>(e);
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, r));
}
__device__ __forceinline__ double cr_cos(double x) {
    const double q = rint(x * 6.36619772367581382433e-01);
    double r = fma(-q, 1.57079632679489655800e+00, x);
    r = fma(-q, 6.12323399573676603587e-17, r);
    const int n = static_cast<int>(q) & 3;
    const double v = (n & 1) ? cr_sin_core(r) : cr_cos_core(r);
    return (n == 1 || n == 2) ? -v : v;
}

__global__ void cr_noise_kernel(float* __restrict__ out, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long m = 2147483647ULL;
    unsigned long long seed = 18101995ULL, pw = 16807ULL;  // seed * a^(2i) mod m by square-and-multiply
    for (unsigned long long k = 2ULL * i; k; k >>= 1) {
        if (k & 1ULL) seed = (seed * pw) % m;
        pw = (pw * pw) % m;
    }
    seed = (16807ULL * seed) % m; const double u1 = __ddiv_rn(static_cast<double>(seed), static_cast<double>(m));
    seed = (16807ULL * seed) % m; const double u2 = __ddiv_rn(static_cast<double>(seed), static_cast<double>(m));
    const double g = __dsqrt_rn(-2.0 * cr_log(u1)) * cr_cos((2.0 * 3.14159265359) * u2);
    out[i] = static_cast<float>(g);
}

hipError_t launch_cr_noise(float* out, size_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(cr_noise_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, out, n);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Integer planes of an intra frame (decode.py:191-206 + png.py:57-58 / yuv.py:152-160).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float round_to_grid(float x, float maxv) { return rintf(maxv * x) / maxv; }

__device__ __forceinline__ unsigned quantise_sample(float x, float maxv) {
    float q = round_to_grid(x, maxv);
    q = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    q = rintf(q * maxv) / maxv;
    return static_cast<unsigned>(rintf(q * maxv));
}

template <typename T>
__global__ void planes_kernel(const float* __restrict__ src, T* p0, T* p1, T* p2, int h, int w, int yuv420, float maxv) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t plane = static_cast<size_t>(h) * w;
    const size_t i = static_cast<size_t>(y) * w + x;
    p0[i] = static_cast<T>(quantise_sample(src[i], maxv));
    if (!yuv420) {
        p1[i] = static_cast<T>(quantise_sample(src[plane + i], maxv));
        p2[i] = static_cast<T>(quantise_sample(src[2 * plane + i], maxv));
        return;
    }
    const int ch = h / 2, cw = w / 2;  // F.avg_pool2d(kernel 2, stride 2), yuv.py:295
    if (y < ch && x < cw) {
        for (int c = 1; c < 3; ++c) {
            const float* s = src + c * plane;
            float sum = 0.0f;  // sequential f32 sum over the 2x2 window of samples already on the bit-depth grid
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) sum += round_to_grid(s[static_cast<size_t>(2 * y + dy) * w + 2 * x + dx], maxv);
            float a = sum / 4.0f;
            a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
            a = rintf(a * maxv) / maxv;
            T* dst = c == 1 ? p1 : p2;
            dst[static_cast<size_t>(y) * cw + x] = static_cast<T>(static_cast<unsigned>(rintf(a * maxv)));
        }
    }
}

hipError_t launch_planes(const float* src, void* p0, void* p1, void* p2, int h, int w, int bitdepth, int frame_data_type,
                         hipStream_t stream) {
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    const float maxv = static_cast<float>((1 << bitdepth) - 1);
    const int yuv420 = frame_data_type == 1;
    if (bitdepth == 8)
        hipLaunchKernelGGL(planes_kernel<uint8_t>, grid, dim3(256), 0, stream, src, static_cast<uint8_t*>(p0),
                           static_cast<uint8_t*>(p1), static_cast<uint8_t*>(p2), h, w, yuv420, maxv);
    else
        hipLaunchKernelGGL(planes_kernel<uint16_t>, grid, dim3(256), 0, stream, src, static_cast<uint16_t*>(p0),
                           static_cast<uint16_t*>(p1), static_cast<uint16_t*>(p2), h, w, yuv420, maxv);
    return hipGetLastError();
}

// u8 planes -> u16 (ccd_decode_video hands out u16 samples whatever the bit depth): done on the device so that the host
// receives the final layout in one copy
__global__ void widen_u8_kernel(const uint8_t* in, uint16_t* out, size_t n) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) out[i] = in[i];
}
hipError_t launch_widen_u8(const uint8_t* in, uint16_t* out, size_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    const unsigned blocks = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(widen_u8_kernel, dim3(blocks), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

}  // namespace ccd
