// ccd_fused_cr.hip - the common-randomness instantiations of the fused float kernel (ccd_fused_kernel.inc, NZ = CIN, kFdPre): the
// synthesis reads the latent levels AND one noise plane per level (bitstream/component/coolchic.py:175-192, component/core/noise.py:
// 17-54; the planes themselves come from ccd_float.hip::cr_noise_kernel + the bicubic x2 chain in stage 1).  Three output
// channels (pictures), 5 .. 9 levels.  A translation unit of its own: compiles next to the other two.
#include "ccd_fused_kernel.inc"

namespace ccd {

template <int CIN>
static hipError_t launch_fdcr(const FusedDec* d_frames, const FdWork* d_work, int n_work, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<CIN, 3, kFdPre, CIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((decode_fused_kernel<CIN, 3, kFdPre, CIN>), dim3(n_work), dim3(kFdThreads), lds, stream, d_frames, d_work);
    return hipGetLastError();
}

bool fused_dec_cr_supports(int c_in, int c) { return c_in >= 5 && c_in <= 9 && c == 3; }

hipError_t launch_fused_dec_cr(const FusedDec* d_frames, const void* d_work, int n_work, int c_in, int c, size_t lds_bytes, hipStream_t stream) {
    if (n_work <= 0) return hipSuccess;
    if (c != 3) return hipErrorInvalidValue;
    const FdWork* w = static_cast<const FdWork*>(d_work);
    switch (c_in) {
        case 5: return launch_fdcr<5>(d_frames, w, n_work, lds_bytes, stream);
        case 6: return launch_fdcr<6>(d_frames, w, n_work, lds_bytes, stream);
        case 7: return launch_fdcr<7>(d_frames, w, n_work, lds_bytes, stream);
        case 8: return launch_fdcr<8>(d_frames, w, n_work, lds_bytes, stream);
        case 9: return launch_fdcr<9>(d_frames, w, n_work, lds_bytes, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ccd
