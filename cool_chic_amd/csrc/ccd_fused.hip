// ccd_fused.hip - host side of the fused float kernel (ccd_fused_kernel.inc) and its kFdWhole instantiations (the whole
// pyramid per tile); the kFdPre / kFdPyr instantiations (level-1 stack from a pyramid launch) live in ccd_fused_pre.hip.
#include "ccd_fused_kernel.inc"

namespace ccd {

int fused_dec_profile_pre(unsigned long long* out16, int reset);  // ccd_fused_pre.hip

// ---- host side ---------------------------------------------------------------------------------------------------
int fused_dec_profile(unsigned long long* out16, int reset) {
#ifdef CCD_FD_PROFILE
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fd_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fd_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    unsigned long long pre[16] = {0};  // the counters of the other translation unit's instantiations
    if (fused_dec_profile_pre(pre, reset) == 1)
        for (int i = 0; i < 16; ++i) out16[i] += pre[i];
    return 1;
#else
    (void)out16; (void)reset;
    return 0;
#endif
}

bool fused_dec_supports(int c_in, int c) { return c_in >= 5 && c_in <= 9 && c >= 2 && c <= 5; }

// LDS of the pyramid launch for a frame of n_lv levels (its descriptor has n_lv - 1 levels and no parameter block)
size_t fused_pyr_lds_bytes(int n_lv) { return static_cast<size_t>(fd_layout(n_lv - 1, 2, false).par) * 4; }

// `pre`: the slot runs the kFdPre instantiation (level-1 stack from the pyramid launch): no LDS for the coarse levels
size_t fused_dec_lds_bytes(int n_lv, int c, int n_conv, int n_params, int pre) {
    (void)n_conv;
    return static_cast<size_t>(fd_layout(n_lv, c, pre != 0).par + ((n_params + 3) & ~3)) * 4;
}

void fused_dec_param_shape(int c_in, int c, int* nwv, int* nws, int* nwc, int* nwo) {
    const int ct = (c + 3) / 4;
    *nwv = (c_in + 4 * ct + 15) / 16; *nws = (c_in * ct + 15) / 16; *nwc = (9 * c * ct + 15) / 16; *nwo = (c * ct + 15) / 16;
}

template <int CIN, int C>
static hipError_t launch_fd(const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<CIN, C, kFdWhole>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_fused_kernel<CIN, C, kFdWhole>), dim3(n_work), dim3(kFdThreads), lds, stream, d_frames, d_work);
    return hipGetLastError();
}

template <int CIN>
static hipError_t launch_fd_c(int c, const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    switch (c) {
        case 2: return launch_fd<CIN, 2>(d_frames, d_work, n_work, lds, stream);
        case 3: return launch_fd<CIN, 3>(d_frames, d_work, n_work, lds, stream);
        case 4: return launch_fd<CIN, 4>(d_frames, d_work, n_work, lds, stream);
        case 5: return launch_fd<CIN, 5>(d_frames, d_work, n_work, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fused_dec_pre(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, size_t lds_bytes, hipStream_t stream);  // ccd_fused_pre.hip

// All frames of one launch share (c_in, c, pre); `d_work` lists (frame, first tile, tile count) per workgroup.
hipError_t launch_fused_dec(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, int pre, size_t lds_bytes, hipStream_t stream) {
    if (n_work <= 0) return hipSuccess;
    if (pre) return launch_fused_dec_pre(d_frames, d_work, n_work, c_in, c, lds_bytes, stream);
    const FdWork* w = static_cast<const FdWork*>(d_work);
    switch (c_in) {
        case 5: return launch_fd_c<5>(c, d_frames, w, n_work, lds_bytes, stream);
        case 6: return launch_fd_c<6>(c, d_frames, w, n_work, lds_bytes, stream);
        case 7: return launch_fd_c<7>(c, d_frames, w, n_work, lds_bytes, stream);
        case 8: return launch_fd_c<8>(c, d_frames, w, n_work, lds_bytes, stream);
        case 9: return launch_fd_c<9>(c, d_frames, w, n_work, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ccd
