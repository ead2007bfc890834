// ccd_rate.hip - per-symbol rate under the Laplace model.
//
// Reference: coolchic/component/core/arm.py:448-485 (_laplace_cdf, compute_rate): for every symbol
//   cdf(t) = 0.5 - 0.5 sign(t - mu) expm1(-|t - mu| / scale)
//   rate   = -log2(max(cdf(x + 0.5) - cdf(x - 0.5), 2^-16))
// in float32, elementwise.  This is the encoder-side (continuous) rate estimate, NOT the range coder's exact
// probability (ccd_entropy*.hip); it is offered because rate-distortion decisions on already-decoded or
// candidate latents use it.  One read of x / mu / scale (12 B) and one write (4 B) per symbol: HBM-bound
// (16 B per symbol algorithmic; bench.py leg `rate_model` reports the achieved GB/s against the 8 TB/s roof).
// Parity: the REFERENCE's compute_rate on 2^16 symbols (tests/golden/rate.npz, tests/golden/gen/dump_rate.py), within 2e-6
// relative + 2e-6 bits + 3e-7 / p bits per symbol (the last term is the formula's own conditioning: p is a difference of
// two CDF values; tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ccd.h"

namespace {

// expm1(t) for t <= 0 in ~12 instructions (the device library's is ~35, and the kernel must stay under the HBM roof's
// instruction budget: 16 B per symbol at 8 TB/s leaves ~75 VALU instructions per symbol): Taylor to degree 7 for t > -0.25
// (remainder 2e-9 relative), exp2(t log2 e) - 1 on the transcendental unit below (absolute error <= 7e-8: the quantity that
// matters, since the caller forms 0.5 -+ 0.5 expm1).  Both are evaluated and one selected: no divergence.
__device__ __forceinline__ float expm1_neg(float t) {
    float q = 1.0f / 5040.0f;
    q = __builtin_fmaf(q, t, 1.0f / 720.0f);
    q = __builtin_fmaf(q, t, 1.0f / 120.0f);
    q = __builtin_fmaf(q, t, 1.0f / 24.0f);
    q = __builtin_fmaf(q, t, 1.0f / 6.0f);
    q = __builtin_fmaf(q, t, 0.5f);
    q = __builtin_fmaf(q, t, 1.0f);
    const float small = q * t;
    const float big = __builtin_amdgcn_exp2f(t * 1.44269504088896340736f) - 1.0f;
    return t > -0.25f ? small : big;
}

// rcp = 1 / scale to within an ulp (v_rcp_f32 + one Newton step): the reference divides, |d| / scale
__device__ __forceinline__ float laplace_cdf(float t, float mu, float rcp) {
    const float d = t - mu;
    const float e = expm1_neg(-fabsf(d) * rcp);
    // sign(d) e with e <= 0 is e carrying the sign of -d (sign(0) = 0 in the reference: e is 0 there anyway)
    return __builtin_fmaf(-0.5f, __builtin_copysignf(e, -d), 0.5f);
}

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1)))* gcv4_t;
typedef v4f __attribute__((address_space(1)))* gv4_t;

__device__ __forceinline__ float symbol_rate(float xi, float m, float s) {
    float r = __builtin_amdgcn_rcpf(s);
    r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
    float p = laplace_cdf(xi + 0.5f, m, r) - laplace_cdf(xi - 0.5f, m, r);
    p = fmaxf(p, 1.52587890625e-05f);  // 2^-16: no symbol costs more than 16 bits
    return -__builtin_amdgcn_logf(p);  // v_log_f32 = log2; p is a normal number in [2^-16, 1]
}

// 16 B/symbol of HBM traffic and ~45 VALU instructions: HBM-bound.  Every lane moves 16 bytes per array and iteration
// (global_load_dwordx4 / global_store_dwordx4: one wave = 1 KB contiguous per array), two iterations' loads in flight
// (8 symbols per lane) before the first use; the grid is a few workgroups per CU and strides over the arrays.
// `n4` = number of whole 4-symbol groups; the (< 4) symbols behind them and unaligned arrays take rate_tail_kernel.
template <bool TOTAL>
__global__ __launch_bounds__(256) void rate_kernel_v4(const float* x_, const float* mu_, const float* scale_, float* rate_, double* total,
                                                      int64_t n4) {
    __shared__ double s_part[4];
    gcv4_t x = (gcv4_t)x_, mu = (gcv4_t)mu_, scale = (gcv4_t)scale_;
    gv4_t rate = (gv4_t)rate_;
    double acc = 0.0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const v4f xa = x[i], ma = mu[i], sa = scale[i];
        const v4f xb = x[i + stride], mb = mu[i + stride], sb = scale[i + stride];
        v4f ra, rb;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ra[k] = symbol_rate(xa[k], ma[k], sa[k]); rb[k] = symbol_rate(xb[k], mb[k], sb[k]); }
        // (streaming stores: the output is not read again by this kernel, it should not displace the inputs' lines)
        if (rate) { __builtin_nontemporal_store(ra, &rate[i]); __builtin_nontemporal_store(rb, &rate[i + stride]); }
        if constexpr (TOTAL) {
            acc += static_cast<double>(ra[0]) + static_cast<double>(ra[1]) + static_cast<double>(ra[2]) + static_cast<double>(ra[3]);
            acc += static_cast<double>(rb[0]) + static_cast<double>(rb[1]) + static_cast<double>(rb[2]) + static_cast<double>(rb[3]);
        }
    }
    if (i < n4) {
        const v4f xa = x[i], ma = mu[i], sa = scale[i];
        v4f ra;
#pragma unroll
        for (int k = 0; k < 4; ++k) ra[k] = symbol_rate(xa[k], ma[k], sa[k]);
        if (rate) __builtin_nontemporal_store(ra, &rate[i]);
        if constexpr (TOTAL) acc += static_cast<double>(ra[0]) + static_cast<double>(ra[1]) + static_cast<double>(ra[2]) + static_cast<double>(ra[3]);
    }
    if (!TOTAL || !total) return;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// scalar form: symbols [first, n) - the tail behind the last whole group of four, or everything when an array is not 16-byte aligned
__global__ __launch_bounds__(256) void rate_tail_kernel(const float* x_, const float* mu_, const float* scale_, float* rate_, double* total,
                                                        int64_t first, int64_t n) {
    __shared__ double s_part[4];
    gcf_t x = (gcf_t)x_, mu = (gcf_t)mu_, scale = (gcf_t)scale_;
    gf_t rate = (gf_t)rate_;
    double acc = 0.0;
    for (int64_t i = first + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
        const float r = symbol_rate(x[i], mu[i], scale[i]);
        if (rate) rate[i] = r;
        acc += static_cast<double>(r);
    }
    if (!total) return;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

}  // namespace

extern "C" int ccd_compute_rate(int device, void* stream, const float* x, const float* mu, const float* scale, int64_t n,
                                float* rate, double* total_bits) {
    if (!x || !mu || !scale || n < 0 || (!rate && !total_bits)) return CCD_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return CCD_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (total_bits && hipMemsetAsync(total_bits, 0, sizeof(double), st) != hipSuccess) return CCD_ERR_HIP;
    if (n == 0) return CCD_OK;
    const auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = aligned(x) && aligned(mu) && aligned(scale) && (!rate || aligned(rate));
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4) {
        // 8 workgroups of 256 per CU (2048 of them) cover the chip; two 16-byte groups per lane and iteration
        const int64_t want = (n4 + 511) / 512;
        const unsigned blocks = static_cast<unsigned>(want < 2048 ? want : 2048);
        if (total_bits) hipLaunchKernelGGL(rate_kernel_v4<true>, dim3(blocks), dim3(256), 0, st, x, mu, scale, rate, total_bits, n4);
        else hipLaunchKernelGGL(rate_kernel_v4<false>, dim3(blocks), dim3(256), 0, st, x, mu, scale, rate, total_bits, n4);
    }
    if (4 * n4 < n) {
        const int64_t rest = n - 4 * n4;
        const int64_t want = (rest + 255) / 256;
        const unsigned blocks = static_cast<unsigned>(want < 8192 ? want : 8192);
        hipLaunchKernelGGL(rate_tail_kernel, dim3(blocks), dim3(256), 0, st, x, mu, scale, rate, total_bits, 4 * n4, n);
    }
    return hipGetLastError() == hipSuccess ? CCD_OK : CCD_ERR_HIP;
}
