// ccd_inter.hip - P / B frame reconstruction on the device (SURVEY.md section 8f "next-1").
//
// Reference behaviour (paths relative to /root/reference/coolchic):
//   bitstream/decode.py:156-189                      global shift, warp, alpha / beta blend, + residue
//   component/intercoding/globalmotion.py:151-160    integer global translation (nearest, border clamp)
//   component/intercoding/warp.py:226-243,294-397    sinc-windowed N-tap warp, TRAINING mode (the decoder never
//                                                    calls .eval(): flows are NOT quantised), border clamp;
//                                                    2 / 4 taps = F.grid_sample bilinear / bicubic (warp.py:325-343)
//   io/format/yuv.py:303-316                         4:2:0 references -> 4:4:4 by nearest x2
//
// Numerics: identical formulas to oracle/cc_oracle.c section 11 (sin / cos in f64 with explicit fma, rounded to
// f32; separable passes accumulated with fmaf, taps ascending) -> bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include "ccd_device.hpp"

namespace ccd {

__device__ __forceinline__ int ic_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double ic_sin_core(double r) {
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
__device__ __forceinline__ double ic_cos_core(double r) {
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
__device__ __forceinline__ void ic_sincos(float a, float* s_out, float* c_out) {
    const double x = static_cast<double>(a);
    const double q = rint(x * 6.36619772367581382433e-01);
    double r = fma(-q, 1.57079632679489655800e+00, x);
    r = fma(-q, 6.12323399573676603587e-17, r);
    const int n = static_cast<int>(q) & 3;
    const double sn = ic_sin_core(r), cs = ic_cos_core(r);
    double sv = (n & 1) ? cs : sn, cv = (n & 1) ? sn : cs;
    if (n & 2) sv = -sv;
    if (n == 1 || n == 2) cv = -cv;
    *s_out = static_cast<float>(sv);
    *c_out = static_cast<float>(cv);
}

constexpr int kMaxTaps = 16;

// warp.py:238-243
__device__ __forceinline__ void ic_coeffs(float s, int n_taps, float* coef) {
    const float pi_f = 3.14159265358979323846f;
    for (int j = 0; j < n_taps; ++j) {
        const float d = s - static_cast<float>(j - n_taps / 2 + 1);
        float sn, unused_c, unused_s, win;
        ic_sincos(pi_f * d / static_cast<float>(n_taps), &unused_s, &win);
        float snc = 1.0f;
        if (d != 0.0f) { const float a = pi_f * d; ic_sincos(a, &sn, &unused_c); snc = sn / a; }
        coef[j] = win * snc;
    }
}

// Integer planes of a decoded frame -> the [3][H][W] float tensor the warper reads (value = q / (2^bd - 1),
// 4:2:0 chroma repeated 2x2).
template <typename T>
__global__ void planes_to_444_kernel(const T* p0, const T* p1, const T* p2, float* out, int h, int w, int chroma_half, float maxv) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t plane = static_cast<size_t>(h) * w, i = static_cast<size_t>(y) * w + x;
    out[i] = static_cast<float>(p0[i]) / maxv;
    const int cw = chroma_half ? w / 2 : w;
    const size_t ci = chroma_half ? static_cast<size_t>(y >> 1) * cw + (x >> 1) : i;
    out[plane + i] = static_cast<float>(p1[ci]) / maxv;
    out[2 * plane + i] = static_cast<float>(p2[ci]) / maxv;
}

hipError_t launch_planes_to_444(const void* p0, const void* p1, const void* p2, float* out, int h, int w, int bitdepth,
                                int frame_data_type, hipStream_t stream) {
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    const float maxv = static_cast<float>((1 << bitdepth) - 1);
    const int half = frame_data_type == 1;
    if (bitdepth == 8)
        hipLaunchKernelGGL(planes_to_444_kernel<uint8_t>, grid, dim3(256), 0, stream, static_cast<const uint8_t*>(p0),
                           static_cast<const uint8_t*>(p1), static_cast<const uint8_t*>(p2), out, h, w, half, maxv);
    else
        hipLaunchKernelGGL(planes_to_444_kernel<uint16_t>, grid, dim3(256), 0, stream, static_cast<const uint16_t*>(p0),
                           static_cast<const uint16_t*>(p1), static_cast<const uint16_t*>(p2), out, h, w, half, maxv);
    return hipGetLastError();
}

__device__ __forceinline__ void ic_warp_pixel(const float* __restrict__ ref, int H, int W, int gx, int gy, int n_taps, float fx,
                                              float fy, int y, int x, float out[3]) {
    const float rxf = floorf(fx), ryf = floorf(fy);
    const float sx = fx - rxf, sy = fy - ryf;
    const int rx = static_cast<int>(rxf), ry = static_cast<int>(ryf);
    float cx[kMaxTaps], cy[kMaxTaps];
    ic_coeffs(sx, n_taps, cx);
    ic_coeffs(sy, n_taps, cy);
    const int lo = -(n_taps / 2) + 1;
    const size_t plane = static_cast<size_t>(H) * W;
    int xs[kMaxTaps];
    for (int j = 0; j < n_taps; ++j) xs[j] = ic_clamp(ic_clamp(x + lo + j + rx, 0, W - 1) + gx, 0, W - 1);
    for (int c = 0; c < 3; ++c) {
        float acc = 0.0f;
        for (int i = 0; i < n_taps; ++i) {
            const int yy = ic_clamp(ic_clamp(y + lo + i + ry, 0, H - 1) + gy, 0, H - 1);
            const float* row = ref + c * plane + static_cast<size_t>(yy) * W;
            float line = 0.0f;
            for (int j = 0; j < n_taps; ++j) line = __fmaf_rn(row[xs[j]], cx[j], line);
            acc = __fmaf_rn(line, cy[i], acc);
        }
        out[c] = acc;
    }
}

// The same with a compile-time tap count (r06): the runtime-sized version keeps cx / cy / xs in scratch memory (dynamic indexing
// of per-thread arrays) and spent 0.5 ms per 1080p frame there - ccd_decode_video's 31 inter frames were 21 ms of a 190 ms call
// (profiles/r06/gop_timing_before.txt).  Same operations in the same order: bit-identical.
// (r06, measured and dropped: evaluating only the polynomial the quadrant needs where a wave agrees on it - the window's cos does,
// per tap - behind a ballot: bit-exact, and 0.55 ms per 1080p frame instead of 0.34: the scalar branches serialise what the
// scheduler interleaves when all 32 evaluations of a pixel are straight-line code.)
template <int NT>
__device__ __forceinline__ void ic_coeffs_t(float s, float (&coef)[NT]) {
    const float pi_f = 3.14159265358979323846f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const float d = s - static_cast<float>(j - NT / 2 + 1);
        float sn, unused_c, unused_s, win;
        ic_sincos(pi_f * d / static_cast<float>(NT), &unused_s, &win);
        float snc = 1.0f;
        if (d != 0.0f) { const float a = pi_f * d; ic_sincos(a, &sn, &unused_c); snc = sn / a; }
        coef[j] = win * snc;
    }
}
// the N x N taps of one pixel from its coefficients (rows accumulate in ascending order per channel: the oracle's chain)
template <int NT>
__device__ __forceinline__ void ic_taps_t(const float* __restrict__ ref, int H, int W, int gx, int gy, int rx, int ry, const float (&cx)[NT],
                                          const float (&cy)[NT], int y, int x, float out[3]) {
    constexpr int lo = -(NT / 2) + 1;
    const size_t plane = static_cast<size_t>(H) * W;
    int xs[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) xs[j] = ic_clamp(ic_clamp(x + lo + j + rx, 0, W - 1) + gx, 0, W - 1);
    float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int yy = ic_clamp(ic_clamp(y + lo + i + ry, 0, H - 1) + gy, 0, H - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* row = ref + c * plane + static_cast<size_t>(yy) * W;
            float line = 0.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j) line = __fmaf_rn(row[xs[j]], cx[j], line);
            acc[c] = __fmaf_rn(line, cy[i], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = acc[c];
}
template <int NT>
__device__ __forceinline__ void ic_warp_pixel_t(const float* __restrict__ ref, int H, int W, int gx, int gy, float fx, float fy, int y, int x, float out[3]) {
    const float rxf = floorf(fx), ryf = floorf(fy);
    const float sx = fx - rxf, sy = fy - ryf;
    float cx[NT], cy[NT];
    ic_coeffs_t<NT>(sx, cx);
    ic_coeffs_t<NT>(sy, cy);
    ic_taps_t<NT>(ref, H, W, gx, gy, static_cast<int>(rxf), static_cast<int>(ryf), cx, cy, y, x, out);
}

// ---- warp_filter_size 2 / 4: the Warper's native path = F.grid_sample(bilinear | bicubic, border, align_corners=True)
// (warp.py:92-116, 325-343).  Same float32 operation sequence as oracle/cc_oracle.c section 11 (the canon of the
// reference run: fused linspace, plain weight products + fma accumulation for bilinear, the mixed plain / fused
// evaluation of the bicubic coefficients and sums).
__device__ __forceinline__ float ic_lin_coord(int i, int n) {
    if (n == 1) return -1.0f;
    const float step = __fdiv_rn(2.0f, static_cast<float>(n - 1));
    return i < n / 2 ? __fmaf_rn(step, static_cast<float>(i), -1.0f) : __fmaf_rn(-step, static_cast<float>(n - 1 - i), 1.0f);
}
__device__ __forceinline__ float ic_cubic_inner(float x) {
    const float A = -0.75f;
    const float t = __fmul_rn(__fmaf_rn(A + 2.0f, x, -(A + 3.0f)), x);
    return __fmaf_rn(t, x, 1.0f);
}
__device__ __forceinline__ float ic_cubic_outer(float x) {
    const float A = -0.75f;
    float t = __fmul_rn(A, x);
    t = __fsub_rn(t, 5.0f * A);
    t = __fmul_rn(t, x);
    t = __fadd_rn(t, 8.0f * A);
    t = __fmul_rn(t, x);
    return __fsub_rn(t, 4.0f * A);
}
__device__ __forceinline__ void ic_warp_pixel_native(const float* __restrict__ ref, int H, int W, int gx, int gy, int n_taps, float fx,
                                                     float fy, int y, int x, float out[3]) {
    const float sx = static_cast<float>((W - 1.0) / 2.0), sy = static_cast<float>((H - 1.0) / 2.0);
    const float gxn = __fadd_rn(ic_lin_coord(x, W), __fdiv_rn(fx, sx)), gyn = __fadd_rn(ic_lin_coord(y, H), __fdiv_rn(fy, sy));
    float ix = __fmul_rn(__fadd_rn(gxn, 1.0f), sx), iy = __fmul_rn(__fadd_rn(gyn, 1.0f), sy);
    const size_t plane = static_cast<size_t>(H) * W;
    if (n_taps == 2) {
        ix = fminf(static_cast<float>(W - 1), fmaxf(ix, 0.0f));
        iy = fminf(static_cast<float>(H - 1), fmaxf(iy, 0.0f));
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float w = __fsub_rn(ix, x0f), e = __fsub_rn(1.0f, w), n = __fsub_rn(iy, y0f), s = __fsub_rn(1.0f, n);
        const float w_nw = __fmul_rn(s, e), w_ne = __fmul_rn(s, w), w_sw = __fmul_rn(n, e), w_se = __fmul_rn(n, w);
        const int x0 = static_cast<int>(x0f), y0 = static_cast<int>(y0f);
        const int xa = ic_clamp(ic_clamp(x0, 0, W - 1) + gx, 0, W - 1), xb = ic_clamp(ic_clamp(x0 + 1, 0, W - 1) + gx, 0, W - 1);
        const int ya = ic_clamp(ic_clamp(y0, 0, H - 1) + gy, 0, H - 1), yb = ic_clamp(ic_clamp(y0 + 1, 0, H - 1) + gy, 0, H - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* r = ref + c * plane;
            float acc = __fmul_rn(r[static_cast<size_t>(ya) * W + xa], w_nw);
            acc = __fmaf_rn(r[static_cast<size_t>(ya) * W + xb], w_ne, acc);
            acc = __fmaf_rn(r[static_cast<size_t>(yb) * W + xa], w_sw, acc);
            acc = __fmaf_rn(r[static_cast<size_t>(yb) * W + xb], w_se, acc);
            out[c] = acc;
        }
        return;
    }
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = __fsub_rn(ix, x0f), ty = __fsub_rn(iy, y0f);
    const float cx[4] = {ic_cubic_outer(__fadd_rn(tx, 1.0f)), ic_cubic_inner(tx), ic_cubic_inner(__fsub_rn(1.0f, tx)), ic_cubic_outer(__fsub_rn(2.0f, tx))};
    const float cy[4] = {ic_cubic_outer(__fadd_rn(ty, 1.0f)), ic_cubic_inner(ty), ic_cubic_inner(__fsub_rn(1.0f, ty)), ic_cubic_outer(__fsub_rn(2.0f, ty))};
    const int x0 = static_cast<int>(fminf(fmaxf(x0f, -4.0f), static_cast<float>(W) + 4.0f));
    const int y0 = static_cast<int>(fminf(fmaxf(y0f, -4.0f), static_cast<float>(H) + 4.0f));
    int xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = ic_clamp(ic_clamp(x0 - 1 + j, 0, W - 1) + gx, 0, W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float row[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = ic_clamp(ic_clamp(y0 - 1 + i, 0, H - 1) + gy, 0, H - 1);
            const float* r = ref + c * plane + static_cast<size_t>(yy) * W;
            float acc = __fmaf_rn(r[xs[0]], cx[0], __fmul_rn(r[xs[1]], cx[1]));
            acc = __fadd_rn(acc, __fmul_rn(r[xs[2]], cx[2]));
            acc = __fadd_rn(acc, __fmul_rn(r[xs[3]], cx[3]));
            row[i] = acc;
        }
        float acc = __fmaf_rn(row[1], cy[1], __fmul_rn(row[0], cy[0]));
        acc = __fmaf_rn(row[2], cy[2], acc);
        acc = __fmaf_rn(row[3], cy[3], acc);
        out[c] = acc;
    }
}

struct InterParams {
    const float* residue;  // [4 | 5][H][W]
    const float* motion;   // [2 | 4][H][W]
    const float* ref0;     // [3][H][W]
    const float* ref1;     // B frames
    float* out;            // [3][H][W]
    int frame_type, H, W, n_taps;
    int gflow[4];
};

template <int NT>  // 0: run-time tap count (any even size, and the native 2- / 4-tap paths); 8: the sinc-8 warp of every preset
__global__ __launch_bounds__(256) void inter_recon_kernel(InterParams p) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.W || y >= p.H) return;
    const size_t plane = static_cast<size_t>(p.H) * p.W, i = static_cast<size_t>(y) * p.W + x;
    float a = p.residue[3 * plane + i] + 0.5f;
    a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
    float w0[3], pred[3];
    if constexpr (NT > 0) ic_warp_pixel_t<NT>(p.ref0, p.H, p.W, p.gflow[0], p.gflow[1], p.motion[i], p.motion[plane + i], y, x, w0);
    else if (p.n_taps < 6) ic_warp_pixel_native(p.ref0, p.H, p.W, p.gflow[0], p.gflow[1], p.n_taps, p.motion[i], p.motion[plane + i], y, x, w0);
    else ic_warp_pixel(p.ref0, p.H, p.W, p.gflow[0], p.gflow[1], p.n_taps, p.motion[i], p.motion[plane + i], y, x, w0);
    if (p.frame_type == 2) {
        float b = p.residue[4 * plane + i] + 0.5f;
        b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
        float w1[3];
        if constexpr (NT > 0) ic_warp_pixel_t<NT>(p.ref1, p.H, p.W, p.gflow[2], p.gflow[3], p.motion[2 * plane + i], p.motion[3 * plane + i], y, x, w1);
        else if (p.n_taps < 6) ic_warp_pixel_native(p.ref1, p.H, p.W, p.gflow[2], p.gflow[3], p.n_taps, p.motion[2 * plane + i], p.motion[3 * plane + i], y, x, w1);
        else ic_warp_pixel(p.ref1, p.H, p.W, p.gflow[2], p.gflow[3], p.n_taps, p.motion[2 * plane + i], p.motion[3 * plane + i], y, x, w1);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float t0 = b * w0[c], t1 = (1.0f - b) * w1[c]; pred[c] = t0 + t1; }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) pred[c] = w0[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float m = a * pred[c]; p.out[c * plane + i] = m + p.residue[c * plane + i]; }
}

// ---- r06: the warp's coefficients ahead of the references.  A frame's flows (the motion cool-chic's output) are known long before
// its references are: in a hierarchical GOP every B-frame cool-chic is decoded at ~0.5 of the I frames' chains (fewer symbols), and
// everything then waits for the I frames.  The f64 sin / cos of the sinc window - 64 evaluations per pixel and reference pair - only
// depend on the flows, so ccd_decode_video CAN compute them in that gap (CCD_VIDEO_COEF_MB: inter_coef8_kernel, 16 coefficients per
// pixel and reference, 64 B) and only gather behind the references (inter_apply8_kernel).  Same functions, same operation order:
// bit-identical to the one-kernel form.  Measured: the gather alone is 0.26 of the 0.34 ms - off by default (ccd_api.cpp).
// coef layout: float4 [ref][4][H * W] - quad 0 / 1 = cx[0..3] / cx[4..7], quad 2 / 3 = cy: every access a coalesced 16 bytes per lane.
__global__ __launch_bounds__(256) void inter_coef8_kernel(const float* __restrict__ motion, int H, int W, int n_refs, float4* __restrict__ coef) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W, i = static_cast<size_t>(y) * W + x;
    for (int r = 0; r < n_refs; ++r) {
        const float fx = motion[(2 * r) * plane + i], fy = motion[(2 * r + 1) * plane + i];
        const float sx = fx - floorf(fx), sy = fy - floorf(fy);
        float cx[8], cy[8];
        ic_coeffs_t<8>(sx, cx);
        ic_coeffs_t<8>(sy, cy);
        float4* o = coef + static_cast<size_t>(r) * 4 * plane + i;
        o[0] = make_float4(cx[0], cx[1], cx[2], cx[3]);
        o[plane] = make_float4(cx[4], cx[5], cx[6], cx[7]);
        o[2 * plane] = make_float4(cy[0], cy[1], cy[2], cy[3]);
        o[3 * plane] = make_float4(cy[4], cy[5], cy[6], cy[7]);
    }
}
__device__ __forceinline__ void ic_warp_pixel_coef8(const float* __restrict__ ref, const float4* __restrict__ coef, size_t plane, size_t i, int H, int W, int gx,
                                                    int gy, float fx, float fy, int y, int x, float out[3]) {
    const float4 a = coef[i], b = coef[plane + i], c = coef[2 * plane + i], d = coef[3 * plane + i];
    const float cx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, cy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    ic_taps_t<8>(ref, H, W, gx, gy, static_cast<int>(floorf(fx)), static_cast<int>(floorf(fy)), cx, cy, y, x, out);
}
__global__ __launch_bounds__(256) void inter_apply8_kernel(InterParams p, const float4* __restrict__ coef) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.W || y >= p.H) return;
    const size_t plane = static_cast<size_t>(p.H) * p.W, i = static_cast<size_t>(y) * p.W + x;
    float a = p.residue[3 * plane + i] + 0.5f;
    a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
    float w0[3], pred[3];
    ic_warp_pixel_coef8(p.ref0, coef, plane, i, p.H, p.W, p.gflow[0], p.gflow[1], p.motion[i], p.motion[plane + i], y, x, w0);
    if (p.frame_type == 2) {
        float b = p.residue[4 * plane + i] + 0.5f;
        b = b < 0.0f ? 0.0f : (b > 1.0f ? 1.0f : b);
        float w1[3];
        ic_warp_pixel_coef8(p.ref1, coef + 4 * plane, plane, i, p.H, p.W, p.gflow[2], p.gflow[3], p.motion[2 * plane + i], p.motion[3 * plane + i], y, x, w1);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float t0 = b * w0[c], t1 = (1.0f - b) * w1[c]; pred[c] = t0 + t1; }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) pred[c] = w0[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float m = a * pred[c]; p.out[c * plane + i] = m + p.residue[c * plane + i]; }
}
size_t inter_coef_bytes(int frame_type, int h, int w) { return static_cast<size_t>(frame_type == 2 ? 2 : 1) * 4 * h * w * sizeof(float4); }
hipError_t launch_inter_coef8(int frame_type, int h, int w, const float* motion, void* coef, hipStream_t stream) {
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(inter_coef8_kernel, grid, dim3(256), 0, stream, motion, h, w, frame_type == 2 ? 2 : 1, static_cast<float4*>(coef));
    return hipGetLastError();
}
hipError_t launch_inter_apply8(int frame_type, int h, int w, const int* gflow, const float* residue, const float* motion, const float* ref0,
                               const float* ref1, const void* coef, float* out, hipStream_t stream) {
    InterParams p;
    p.residue = residue; p.motion = motion; p.ref0 = ref0; p.ref1 = ref1; p.out = out;
    p.frame_type = frame_type; p.H = h; p.W = w; p.n_taps = 8;
    for (int i = 0; i < 4; ++i) p.gflow[i] = gflow[i];
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(inter_apply8_kernel, grid, dim3(256), 0, stream, p, static_cast<const float4*>(coef));
    return hipGetLastError();
}

hipError_t launch_inter_recon(int frame_type, int h, int w, int n_taps, const int* gflow, const float* residue, const float* motion,
                              const float* ref0, const float* ref1, float* out, hipStream_t stream) {
    InterParams p;
    p.residue = residue; p.motion = motion; p.ref0 = ref0; p.ref1 = ref1; p.out = out;
    p.frame_type = frame_type; p.H = h; p.W = w; p.n_taps = n_taps;
    for (int i = 0; i < 4; ++i) p.gflow[i] = gflow[i];
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    if (n_taps == 8) hipLaunchKernelGGL(inter_recon_kernel<8>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(inter_recon_kernel<0>, grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---- a kernel that only takes time: ccd_api.cpp measures with it which of the library's side streams run concurrently
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
hipError_t launch_spin(unsigned long long ticks, hipStream_t stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
    return hipGetLastError();
}

}  // namespace ccd
