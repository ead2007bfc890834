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

#include "ccd_device.hpp"

namespace ccd {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// -------------------------------------------------------------------------------------------------
// Upsampling: one launch per pyramid level.  blockIdx.z = output channel: 0 is the pre-concat conv of
// the level's own latent, c >= 1 is the x2 transposed conv of input channel c-1.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_level_kernel(UpsampleLevel L) {
    const int ox = blockIdx.x * 32 + (threadIdx.x & 31);
    const int oy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (ox >= L.w_out || oy >= L.h_out) return;
    const int ch = blockIdx.z;
    float acc = 0.0f;
    if (ch == 0) {
        // pre-concatenation filter: zero padding, kron kernel, residual (upsampling.py:189-196)
        const int k = L.pre_k, pad = k / 2;
        const int8_t* t = L.target;
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
        // x2 transposed conv on the replicate-padded input, cropped by C (upsampling.py:306-325)
        const int k = L.ups_k, p0 = k / 2, crop = 2 * p0 - 1 + k / 2;
        const int h = L.h_in, w = L.w_in;
        const int py = oy + crop, px = ox + crop;
        const float* inf = L.in_f32 ? L.in_f32 + static_cast<size_t>(ch - 1) * h * w : nullptr;
        for (int ky = py & 1; ky < k; ky += 2) {
            const int iy = (py - ky) / 2;
            if (py - ky < 0 || iy >= h + 2 * p0) continue;
            const int sy = clampi(iy - p0, 0, h - 1);
            for (int kx = px & 1; kx < k; kx += 2) {
                const int ix = (px - kx) / 2;
                if (px - kx < 0 || ix >= w + 2 * p0) continue;
                const int sx = clampi(ix - p0, 0, w - 1);
                const float k2 = L.ups_w[ky] * L.ups_w[kx];
                const float v = inf ? inf[sy * w + sx] : static_cast<float>(L.in_i8[sy * w + sx]);
                acc = __fmaf_rn(v, k2, acc);
            }
        }
    }
    L.out[(static_cast<size_t>(ch) * L.h_out + oy) * L.w_out + ox] = acc;
}

hipError_t launch_upsample_level(const UpsampleLevel& L, hipStream_t stream) {
    dim3 grid((L.w_out + 31) / 32, (L.h_out + 7) / 8, L.c_in + 1);
    hipLaunchKernelGGL(upsample_level_kernel, grid, dim3(256), 0, stream, L);
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
        if (relu) acc = acc > 0.0f ? acc : 0.0f;
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

}  // namespace ccd
