// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo's kernels use
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").  Every kernel streams
// exactly N bytes (N = 1 GiB: four times the Infinity Cache) with one access width per lane:
//   read  1 B / lane (the fused float kernel's latent bytes before they were widened, the PNG filters' samples)
//   read  2 B / lane (the entropy kernel's int16 feature planes)
//   read  4 B / lane (dword loads of latent bytes, f32 planes of the unfused path)
//   read 16 B / lane (the guide's calibrated case: FETCH_SIZE reports half)
//   write 1 B / lane (latent grids, 8-bit planes), 4 B, 8 B (partial plane stores of the fused kernel), 16 B
// tools/pmc_calibrate.sh runs it once per counter and writes profiles/r03/pmc_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__global__ void read_k(const T* in, size_t n, T* sink) {
    T acc = T();
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const T v = in[i];
        if constexpr (sizeof(T) == 16) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; } else acc ^= v;
    }
    bool odd;
    if constexpr (sizeof(T) == 16) odd = (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u; else odd = acc == static_cast<T>(0x79);
    if (odd) sink[threadIdx.x] = acc;  // never true for the zero-filled input: keeps the loads alive, writes nothing
}
template <typename T>
__global__ void write_k(T* out, size_t n, T v) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) out[i] = v;
}

int main() {
    const size_t N = size_t{1} << 30;
    void* buf; void* sink;
    if (hipMalloc(&buf, N) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, N);
    (void)hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_k<uint8_t>, grid, block, 0, 0, static_cast<const uint8_t*>(buf), N, static_cast<uint8_t*>(sink));
        hipLaunchKernelGGL(read_k<uint16_t>, grid, block, 0, 0, static_cast<const uint16_t*>(buf), N / 2, static_cast<uint16_t*>(sink));
        hipLaunchKernelGGL(read_k<uint32_t>, grid, block, 0, 0, static_cast<const uint32_t*>(buf), N / 4, static_cast<uint32_t*>(sink));
        hipLaunchKernelGGL(read_k<uint4>, grid, block, 0, 0, static_cast<const uint4*>(buf), N / 16, static_cast<uint4*>(sink));
        hipLaunchKernelGGL(write_k<uint8_t>, grid, block, 0, 0, static_cast<uint8_t*>(buf), N, uint8_t{0});
        hipLaunchKernelGGL(write_k<uint32_t>, grid, block, 0, 0, static_cast<uint32_t*>(buf), N / 4, 0u);
        hipLaunchKernelGGL(write_k<uint2>, grid, block, 0, 0, static_cast<uint2*>(buf), N / 8, make_uint2(0, 0));
        hipLaunchKernelGGL(write_k<uint4>, grid, block, 0, 0, static_cast<uint4*>(buf), N / 16, make_uint4(0, 0, 0, 0));
        (void)hipDeviceSynchronize();
    }
    printf("pmc_calib: every kernel streamed %zu bytes\n", N);
    return 0;
}
