// ccd_rate.hip - per-symbol rate under the Laplace model.
//
// Reference: coolchic/component/core/arm.py:448-485 (_laplace_cdf, compute_rate): for every symbol
//   cdf(t) = 0.5 - 0.5 sign(t - mu) expm1(-|t - mu| / scale)
//   rate   = -log2(max(cdf(x + 0.5) - cdf(x - 0.5), 2^-16))
// in float32, elementwise.  This is the encoder-side (continuous) rate estimate, NOT the range coder's exact
// probability (ccd_entropy*.hip); it is offered because rate-distortion decisions on already-decoded or
// candidate latents use it.  One read of x / mu / scale (12 B) and one write (4 B) per symbol: HBM-bound.
// Parity: the same float32 formula in PyTorch, within 2e-6 relative + 2e-6 bits + 3e-7 / p bits per symbol (the last
// term is the formula's own conditioning: p is a difference of two CDF values; tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ccd.h"

namespace {

__device__ __forceinline__ float laplace_cdf(float t, float mu, float scale) {
    const float d = t - mu;
    const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return 0.5f - 0.5f * sgn * expm1f(-fabsf(d) / scale);
}

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;

__global__ __launch_bounds__(256) void rate_kernel(const float* x_, const float* mu_, const float* scale_, float* rate_, double* total,
                                                   int64_t n) {
    __shared__ double s_part[4];
    gcf_t x = (gcf_t)x_, mu = (gcf_t)mu_, scale = (gcf_t)scale_;
    gf_t rate = (gf_t)rate_;
    double acc = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
        const float xi = x[i], m = mu[i], s = scale[i];
        float p = laplace_cdf(xi + 0.5f, m, s) - laplace_cdf(xi - 0.5f, m, s);
        p = fmaxf(p, 1.52587890625e-05f);  // 2^-16: no symbol costs more than 16 bits
        const float r = -log2f(p);
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
    // enough workgroups to fill 256 CUs several times over; grid-stride beyond that
    const int64_t want = (n + 255) / 256;
    const unsigned blocks = static_cast<unsigned>(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(rate_kernel, dim3(blocks), dim3(256), 0, st, x, mu, scale, rate, total_bits, n);
    return hipGetLastError() == hipSuccess ? CCD_OK : CCD_ERR_HIP;
}
