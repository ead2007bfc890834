// r06: how many launches on different HIP streams really run at once, and do two launches that together need every CU co-run?
// (ccd_batch_run forks the entropy launches of a batch over side streams: profiles/r06/queues.txt decides how many, and when.)
//   1. k spin kernels (1 workgroup, ~5 ms each) on the null stream + k - 1 non-blocking streams created like libccd's
//      (1 upload + 8 side streams): wall time / 5 ms = how many serialised.
//   2. "one workgroup per CU" kernels (139 KB of LDS, 512 threads, like entropy_pipe_kernel): A workgroups on the null stream,
//      B on a side stream, A + B <= 256: wall time against one launch of A + B workgroups.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/queues.hip -o tools/ubench/queues
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void spin_kernel(uint64_t ticks, uint64_t* out) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(32); t = __builtin_amdgcn_s_memtime(); }
    if (threadIdx.x == 0 && out) out[blockIdx.x] = t - t0;
}

extern __shared__ unsigned char big_lds[];
__global__ __launch_bounds__(512) void fat_kernel(uint64_t ticks, uint32_t* where) {
    big_lds[threadIdx.x] = static_cast<unsigned char>(threadIdx.x);
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint64_t t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(32); t = __builtin_amdgcn_s_memtime(); }
    if (threadIdx.x == 0 && where) {
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        where[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);  // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    }
    if (big_lds[(threadIdx.x + 1) & 511] == 255 && ticks == 1) printf("x");
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t up, side[8];
    CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    for (auto& s : side) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // memtime ticks at 100 MHz on gfx950? calibrate: spin for N ticks, measure wall
    const uint64_t probe = 1000000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, probe, nullptr);
    CHECK(hipDeviceSynchronize());
    double t0 = now_ms();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, probe, nullptr);
    CHECK(hipDeviceSynchronize());
    const double ms_per_mtick = now_ms() - t0;
    printf("s_memtime: 1e6 ticks = %.3f ms\n", ms_per_mtick);
    const uint64_t t5 = static_cast<uint64_t>(5.0 / ms_per_mtick * 1e6);
    // ---- 1. concurrency of k launches
    for (int k = 1; k <= 9; ++k) {
        CHECK(hipDeviceSynchronize());
        t0 = now_ms();
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, i == 0 ? nullptr : side[i - 1], t5, nullptr);
        CHECK(hipDeviceSynchronize());
        printf("%d launches of 5 ms on the null stream + %d side streams: %.2f ms wall\n", k, k - 1, now_ms() - t0);
    }
    // pairs: null stream + side[j]
    for (int j = 0; j < 8; ++j) {
        CHECK(hipDeviceSynchronize());
        t0 = now_ms();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, nullptr, t5, nullptr);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, side[j], t5, nullptr);
        CHECK(hipDeviceSynchronize());
        printf("null stream + side[%d]: %.2f ms\n", j, now_ms() - t0);
    }
    // a user stream (like a torch side stream) + side[j]
    hipStream_t user;
    CHECK(hipStreamCreateWithFlags(&user, hipStreamDefault));
    for (int j = 0; j < 8; ++j) {
        CHECK(hipDeviceSynchronize());
        t0 = now_ms();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, user, t5, nullptr);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, side[j], t5, nullptr);
        CHECK(hipDeviceSynchronize());
        printf("user stream + side[%d]: %.2f ms\n", j, now_ms() - t0);
    }
    // ---- 2. one workgroup per CU
    const size_t lds = 139 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    uint32_t* where;
    CHECK(hipMalloc(&where, 512 * 4));
    std::vector<uint32_t> hw(512);
    auto one = [&](int n) -> double {
        (void)hipDeviceSynchronize();
        const double a = now_ms();
        hipLaunchKernelGGL(fat_kernel, dim3(n), dim3(512), lds, nullptr, t5, where);
        (void)hipDeviceSynchronize();
        return now_ms() - a;
    };
    one(8);
    for (int n : {24, 64, 128, 192, 256, 257}) printf("one launch of %3d fat workgroups: %.2f ms\n", n, one(n));
    {   // where do 64 workgroups land?
        one(64);
        CHECK(hipMemcpy(hw.data(), where, 64 * 4, hipMemcpyDeviceToHost));
        int per[8][8] = {};
        for (int i = 0; i < 64; ++i) per[(hw[i] >> 16) & 7][(hw[i] >> 13) & 7]++;
        printf("64 workgroups: per XCC x SE: ");
        for (int x = 0; x < 8; ++x) { for (int s = 0; s < 4; ++s) printf("%d", per[x][s]); printf(" "); }
        printf("\n");
        for (int i = 0; i < 16; ++i) printf("  wg %2d: xcc %u se %u sh %u cu %u\n", i, (hw[i] >> 16) & 0xf, (hw[i] >> 13) & 7, (hw[i] >> 12) & 1, (hw[i] >> 8) & 0xf);
    }
    const int pairs[][2] = {{6, 18}, {64, 64}, {64, 128}, {64, 192}, {192, 64}, {128, 128}, {32, 96}, {100, 100}, {120, 130}};
    for (auto& p : pairs) {
        CHECK(hipDeviceSynchronize());
        t0 = now_ms();
        hipLaunchKernelGGL(fat_kernel, dim3(p[0]), dim3(512), lds, nullptr, t5, nullptr);
        hipLaunchKernelGGL(fat_kernel, dim3(p[1]), dim3(512), lds, side[0], t5, nullptr);
        CHECK(hipDeviceSynchronize());
        printf("two launches %3d (null stream) + %3d (side[0]) fat workgroups: %.2f ms\n", p[0], p[1], now_ms() - t0);
    }
    for (int k = 2; k <= 5; ++k) {  // k launches of 40 fat workgroups each
        CHECK(hipDeviceSynchronize());
        t0 = now_ms();
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(fat_kernel, dim3(40), dim3(512), lds, i == 0 ? nullptr : side[i - 1], t5, nullptr);
        CHECK(hipDeviceSynchronize());
        printf("%d launches of 40 fat workgroups: %.2f ms\n", k, now_ms() - t0);
    }
    return 0;
}
