// ccd_fused_pre.hip - the kFdPre and kFdPyr instantiations of the fused float kernel (ccd_fused_kernel.inc): levels >= 1 of the
// latent pyramid are evaluated once per frame by the pyramid launch (kFdPyr), the tiles of the main launch (kFdPre) load their
// level-1 footprint and run level 0 + synthesis.  A translation unit of its own so that the sets of instantiations compile in
// parallel.
#include "ccd_fused_kernel.inc"

namespace ccd {

int fused_dec_profile_pre(unsigned long long* out16, int reset) {
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

template <int CIN, int C>
static hipError_t launch_fdp(const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<CIN, C, kFdPre>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_fused_kernel<CIN, C, kFdPre>), dim3(n_work), dim3(kFdThreads), lds, stream, d_frames, d_work);
    return hipGetLastError();
}

template <int CIN>
static hipError_t launch_fdp_c(int c, const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    switch (c) {
        case 2: return launch_fdp<CIN, 2>(d_frames, d_work, n_work, lds, stream);
        case 3: return launch_fdp<CIN, 3>(d_frames, d_work, n_work, lds, stream);
        case 4: return launch_fdp<CIN, 4>(d_frames, d_work, n_work, lds, stream);
        case 5: return launch_fdp<CIN, 5>(d_frames, d_work, n_work, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fused_dec_pre(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, size_t lds_bytes, hipStream_t stream) {
    const FdWork* w = static_cast<const FdWork*>(d_work);
    switch (c_in) {
        case 5: return launch_fdp_c<5>(c, d_frames, w, n_work, lds_bytes, stream);
        case 6: return launch_fdp_c<6>(c, d_frames, w, n_work, lds_bytes, stream);
        case 7: return launch_fdp_c<7>(c, d_frames, w, n_work, lds_bytes, stream);
        case 8: return launch_fdp_c<8>(c, d_frames, w, n_work, lds_bytes, stream);
        case 9: return launch_fdp_c<9>(c, d_frames, w, n_work, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

// ---- the pyramid launch: frames of n_lv levels -> descriptors of n_lv - 1 levels (level 0 = the frame's level 1) ---------------
template <int CIN>
static hipError_t launch_pyr(const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<CIN, 2, kFdPyr>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_fused_kernel<CIN, 2, kFdPyr>), dim3(n_work), dim3(kFdThreads), lds, stream, d_frames, d_work);
    return hipGetLastError();
}

// `levels` = levels of the DESCRIPTORS (frame levels - 1): 4 .. 8
hipError_t launch_fused_pyramid(const FusedDec* d_frames, const void* d_work, int n_work, int levels, size_t lds_bytes, hipStream_t stream) {
    if (n_work <= 0) return hipSuccess;
    const FdWork* w = static_cast<const FdWork*>(d_work);
    switch (levels) {
        case 4: return launch_pyr<4>(d_frames, w, n_work, lds_bytes, stream);
        case 5: return launch_pyr<5>(d_frames, w, n_work, lds_bytes, stream);
        case 6: return launch_pyr<6>(d_frames, w, n_work, lds_bytes, stream);
        case 7: return launch_pyr<7>(d_frames, w, n_work, lds_bytes, stream);
        case 8: return launch_pyr<8>(d_frames, w, n_work, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ccd
