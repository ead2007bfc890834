// ccd_entropy.hip - the serial heart of the decoder: integer ARM/IFCE entropy model walked in
// wavefront order + the range decoder, one workgroup per cool-chic, many cool-chics per launch.
//
// Reference behaviour restated here (paths relative to /root/reference/coolchic):
//   bitstream/component/coolchic.py:89-169   per-grid loop, IFCE on the nearest-upsampled stack
//   bitstream/component/latent.py:18-187     wavefront order x + 10 y, context gather, scatter
//   bitstream/component/armint.py:180-203    fixed-point MLP (int64, wrap-around)
//   bitstream/component/rangecoder.py:80-94  -> constriction 0.4.2 RangeDecoder + QuantizedLaplace(-64,63)
//
// Structure (v1, barrier-phased): for every wavefront step the workgroup
//   A. gathers the contexts of the step's pixels and runs the MLP, one work item per
//      (pixel, output neuron), activations exchanged through LDS;
//   B. expands each pixel's (mu, scale) into the 128 left-cumulatives of the leaky quantised
//      Laplace model (f64 exp), one work item per (pixel, symbol);
//   C. wave 0 advances the range decoder pixel by pixel: every lane multiplies two candidate
//      cumulatives by the coder's scale and one ballot/popcount finds the symbol - no division,
//      no search loop on the dependent chain.
#include <hip/hip_runtime.h>

#include "ccd_device.hpp"
#include "ccd_laplace.hpp"

namespace ccd {

constexpr int kEntThreads = 1024;  // 4 waves per SIMD: every phase is a handful of LDS / L2 round trips per item, and only other waves hide them (r05: 256 -> 1024)
constexpr int kChunk = 64;  // pixels of one wavefront step handled per phase round

// Left cumulative of symbol s (> -64) under table indices (mu_idx, scale index): window_left of ccd_laplace.hpp, the function the
// pipelined kernel uses (r05; r01-r04 called the device library's exp here: ~3 x the instructions, same 24-bit boundaries - both
// were enumerated against libm, profiles/r03/cdf_sweep.log).  `rcp` = RN(1 / b) from the host's table (EntropyParams::rcp_table).
__device__ __forceinline__ uint32_t laplace_left(int mu_idx, double rcp, int s) {
    const double mu = -64.0 + static_cast<double>(mu_idx) * (1.0 / 256.0);  // float32 table value, exact
    return window_left(mu, rcp, s, kExpTab);
}

template <bool NARROW>
__device__ __forceinline__ uint64_t mac(uint64_t acc, int64_t x, int64_t w) {
    if (NARROW) return acc + static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(x)) * static_cast<int64_t>(static_cast<int32_t>(w)));
    return acc + static_cast<uint64_t>(x) * static_cast<uint64_t>(w);
}

struct EntShared {
    int err;
};

template <bool NARROW>
__device__ void entropy_decode_slot(const EntropyParams& P, unsigned char* smem_raw) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int dim = P.dim;
    // ---- LDS carve-up ---------------------------------------------------------------------------
    int64_t* s_arm = reinterpret_cast<int64_t*>(smem_raw);                    // [arm_len]
    int64_t* s_xa = s_arm + ((P.arm_len + 1) & ~1);                            // [dim][kChunk]
    int64_t* s_xb = s_xa + dim * kChunk;                                      // [dim][kChunk]
    int64_t* s_stab = s_xb + dim * kChunk;                                    // [2][kChunk]
    uint32_t* s_tbl = reinterpret_cast<uint32_t*>(s_stab + 2 * kChunk);       // [kChunk][128]
    int32_t* s_mu = reinterpret_cast<int32_t*>(s_tbl + kChunk * kAlphabet);   // [kChunk]
    int32_t* s_sc = s_mu + kChunk;                                            // [kChunk]
    int32_t* s_py = s_sc + kChunk;                                            // [kChunk]
    int32_t* s_px = s_py + kChunk;                                            // [kChunk]
    int32_t* s_err = s_px + kChunk;                                           // [1] (+ 3 words of padding)
    double* s_rcpx = reinterpret_cast<double*>(s_err + 4);                    // [kChunk] RN(1 / b) of the chunk's pixels

    for (int i = tid; i < P.arm_len; i += kEntThreads) s_arm[i] = P.arm[i];
    if (tid == 0) *s_err = 0;

    // ---- range decoder state (meaningful in wave 0) ---------------------------
    // dist = point - lower is all the decoder ever uses, so track it directly.
    // (the coder's state is the same in every lane of wave 0: said with readfirstlane, it lives in scalar registers and the
    // symbol loop's arithmetic, compares and branches run on the scalar unit - r05; as per-lane values the loop was ~80 vector
    // instructions with exec-mask control flow per symbol)
    auto uni32 = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    auto uni64 = [&](uint64_t v) { return (static_cast<uint64_t>(uni32(static_cast<uint32_t>(v >> 32))) << 32) | uni32(static_cast<uint32_t>(v)); };
    uint32_t word_pos = 0;
    auto next_word = [&]() -> uint32_t {
        const uint32_t w = uni32(word_pos < P.n_words ? P.words[word_pos] : 0u);
        ++word_pos;
        return w;
    };
    uint64_t rc_range = ~uint64_t{0};
    uint64_t rc_dist = static_cast<uint64_t>(next_word()) << 32;
    rc_dist |= next_word();
    uint64_t n_decoded = 0;
#ifdef CCD_GEN_PROFILE
    unsigned long long gp[6] = {0, 0, 0, 0, 0, 0}, gp_t = 0;
#define GP_START() gp_t = __builtin_amdgcn_s_memtime()
#define GP_ADD(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gp[k] += t_ - gp_t; gp_t = t_; } while (0)
#else
#define GP_START() (void)0
#define GP_ADD(k) (void)0
#endif
    __syncthreads();

    const int n_layers = P.n_layers;
    const int n_sp = P.n_spatial;
    const int n_if = P.has_ifce ? P.n_ifce_out : 0;

    for (int g = P.n_grids - 1; g >= 0; --g) {
        const int H = P.grid_h[g], W = P.grid_w[g];
        int8_t* __restrict__ lat = P.latent[g];
        // ---- IFCE features at the size of the previously decoded grid (coolchic.py:94-146) -------
        const int fin = P.ifce_in[g];
        const int fh = (g == P.n_grids - 1) ? H : P.grid_h[g + 1];
        const int fw = (g == P.n_grids - 1) ? W : P.grid_w[g + 1];
        if (fin > 0) {
            const int64_t* fw_ = P.ifce + P.ifce_off[g];  // w[fin][n_if]
            const int64_t* fb_ = fw_ + fin * n_if;
            const int base_level = (g == P.n_grids - 1) ? 0 : P.level[g + 1];
            for (int p = tid; p < fh * fw; p += kEntThreads) {
                const int y = p / fw, x = p - y * fw;
                for (int o = 0; o < n_if; ++o) {
                    uint64_t acc = static_cast<uint64_t>(fb_[o]);
                    if (g != P.n_grids - 1) {
                        for (int c = 0; c < fin; ++c) {
                            const int m = g + 1 + c;
                            const int sh = P.level[m] - base_level;
                            const int64_t v = P.latent[m][(y >> sh) * P.grid_w[m] + (x >> sh)];
                            acc += static_cast<uint64_t>(v << 16) * static_cast<uint64_t>(fw_[c * n_if + o]);
                        }
                    }  // first grid: the stack is a single all-zero channel (coolchic.py:95-96)
                    const int64_t q8 = static_cast<int64_t>(acc) >> 24;
                    // .to(torch.float) / back to int64 round trip around F.interpolate (coolchic.py:142-144)
                    P.ifce_feat[o * fh * fw + p] = static_cast<int32_t>(static_cast<int64_t>(static_cast<float>(q8)));
                }
            }
        }
        __syncthreads();

        // ---- wavefront walk (latent.py:66-140): coding order x + 10 y, raster if W <= 9 -------------
        const bool raster = W <= 9;
        const int n_steps = raster ? H * W : W + 10 * (H - 1);
        for (int c = 0; c < n_steps; ++c) {
            int y0, x0, n;
            if (raster) { y0 = c / W; x0 = c - y0 * W; n = 1; }
            else {
                if (c < W) { y0 = 0; x0 = c; }
                else { y0 = (c - W) / 10 + 1; x0 = W - 10 + (c - W) % 10; }
                n = min(H - y0, x0 / 10 + 1);  // pixels (y0 + i, x0 - 10 i) inside the grid
            }
            for (int i0 = 0; i0 < n; i0 += kChunk) {
                const int cnt = min(kChunk, n - i0);
                GP_START();
                // ---- A1: gather contexts (already << 16, armint.py:193) ---------------------------
                for (int it = tid; it < cnt * dim; it += kEntThreads) {
                    const int k = it / cnt, i = it - k * cnt;
                    const int y = y0 + i0 + i, x = x0 - 10 * (i0 + i);
                    int64_t v = 0;
                    if (k < n_sp) {
                        const int yy = y - P.ctx_dy[k], xx = x + P.ctx_dx[k];
                        if (yy >= 0 && xx >= 0 && xx < W) v = lat[yy * W + xx];
                    } else if (fin > 0) {
                        v = P.ifce_feat[(k - n_sp) * fh * fw + (y >> 1) * fw + (x >> 1)];
                    }
                    s_xa[k * kChunk + i] = v << 16;
                    if (k == 0) { s_py[i] = y; s_px[i] = x; }
                }
                __syncthreads();
                GP_ADD(0);
                // ---- A2: stabiliser branch + hidden layers ---------------------------------------
                {
                    const int64_t* ws = s_arm + (P.arm_len - 2 - 2 * dim);
                    const int64_t* bs = ws + 2 * dim;
                    for (int it = tid; it < cnt * 2; it += kEntThreads) {
                        const int o = it / cnt, i = it - o * cnt;
                        uint64_t acc = static_cast<uint64_t>(bs[o]);
                        // (four operand pairs in flight: one wave per SIMD has nothing else to hide an LDS round trip behind)
#pragma unroll 4
                        for (int k = 0; k < dim; ++k) acc = mac<NARROW>(acc, s_xa[k * kChunk + i], ws[k * 2 + o]);
                        s_stab[o * kChunk + i] = static_cast<int64_t>(acc);
                    }
                }
                int64_t* xin = s_xa;
                int64_t* xout = s_xb;
                const int64_t* lw = s_arm;
                for (int l = 0; l < n_layers - 1; ++l) {
                    const int64_t* lb = lw + dim * dim;
                    for (int it = tid; it < cnt * dim; it += kEntThreads) {
                        const int o = it / cnt, i = it - o * cnt;
                        uint64_t acc = static_cast<uint64_t>(lb[o]);
                        // (four operand pairs in flight: one wave per SIMD has nothing else to hide an LDS round trip behind)
#pragma unroll 4
                        for (int k = 0; k < dim; ++k) acc = mac<NARROW>(acc, xin[k * kChunk + i], lw[k * dim + o]);
                        int64_t v = static_cast<int64_t>(acc);
                        v = v < 0 ? 0 : v;
                        xout[o * kChunk + i] = v >> 16;
                    }
                    __syncthreads();
                    int64_t* t = xin; xin = xout; xout = t;
                    lw = lb + dim;
                }
                if (n_layers == 1) __syncthreads();  // stabiliser results visible
                GP_ADD(1);
                // ---- A3: output layer -> table indices (latent.py:156-165, rangecoder.py:90-91) ----
                {
                    const int64_t* lb = lw + dim * 2;
                    for (int it = tid; it < cnt * 2; it += kEntThreads) {
                        const int o = it / cnt, i = it - o * cnt;
                        uint64_t acc = static_cast<uint64_t>(lb[o]);
                        // (four operand pairs in flight: one wave per SIMD has nothing else to hide an LDS round trip behind)
#pragma unroll 4
                        for (int k = 0; k < dim; ++k) acc = mac<NARROW>(acc, xin[k * kChunk + i], lw[k * 2 + o]);
                        acc += static_cast<uint64_t>(s_stab[o * kChunk + i]);
                        const int64_t q8 = static_cast<int64_t>(acc) >> 24;
                        if (o == 0) {
                            const int64_t idx = q8 + kMuOffset;
                            s_mu[i] = static_cast<int32_t>(idx < 0 ? 0 : (idx > kNumMu - 1 ? kNumMu - 1 : idx));
                        } else {
                            const int64_t idx = q8 + kScaleOffset;
                            s_sc[i] = static_cast<int32_t>(idx < 0 ? 0 : (idx > kNumScale - 1 ? kNumScale - 1 : idx));
                            s_rcpx[i] = P.rcp_table[s_sc[i]];  // one table read per pixel here, not one per boundary in phase B
                        }
                    }
                }
                __syncthreads();
                GP_ADD(2);
                // ---- B: 128 left cumulatives per pixel ---------------------------------------------
                for (int it = tid; it < cnt * kAlphabet; it += kEntThreads) {
                    const int i = it >> 7, j = it & 127;
                    s_tbl[it] = (j == 0) ? 0u : laplace_left(s_mu[i], s_rcpx[i], j + kAcLo);
                }
                __syncthreads();
                GP_ADD(3);
                // ---- C: range decoder, wave 0 only -------------------------------------------------
                if (tid < 64 && *s_err == 0) {
                    // (r05) the next pixel's two table rows are requested one symbol ahead, and the symbol's bounds come out of the
                    // lanes by v_readlane with the (wave-uniform) index in a scalar register - r01-r04 paid two LDS round trips for
                    // the rows and two more for the __shfl pair on every symbol's chain
                    uint32_t l0 = s_tbl[lane], l1 = s_tbl[64 + lane];
                    int sym_l = 0;  // lane i: the symbol of pixel i of the chunk (stored by all lanes at once behind the loop)
                    int n_done = 0;
                    rc_dist = uni64(rc_dist); rc_range = uni64(rc_range);
                    word_pos = uni32(word_pos);
                    const uint32_t tbl_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s_tbl + lane));  // LDS byte address of entry `lane` of row 0
                    for (int i = 0; i < cnt; ++i) {
                        // the next pixel's two rows: requested NOW, first used at the end of this iteration (inline asm: left to the
                        // compiler the reads sink behind the scalar arithmetic and their round trip lands on the symbol's chain)
                        uint32_t n0, n1;
                        {
                            const uint32_t a = tbl_lane + static_cast<uint32_t>(min(i + 1, cnt - 1)) * (kAlphabet * 4u);
                            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256" : "=&v"(n0), "=&v"(n1) : "v"(a) : "memory");
                        }
                        const uint64_t scale = rc_range >> kRcPrecision;
                        // quantile >= 2^24: invalid data.  (Both operands are below 2^40: the sign of their 64-bit difference decides -
                        // scalar subtract and a 32-bit test; a 64-bit "<" only exists on the vector unit.)
                        uint32_t diff_hi = static_cast<uint32_t>(((rc_dist >> kRcPrecision) - scale) >> 32);
                        asm volatile("" : "+s"(diff_hi));  // (opaque: else the test is re-formed as a 64-bit compare, i.e. moved to the vector unit)
                        if (static_cast<int32_t>(diff_hi) >= 0) {
                            if (lane == 0) *s_err = CCD_ERR_INVALID_DATA;
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            break;
                        }
                        const unsigned long long m0 = __ballot(scale * l0 <= rc_dist);
                        const unsigned long long m1 = __ballot(scale * l1 <= rc_dist);
                        const int sidx = __builtin_amdgcn_readfirstlane(__popcll(m0) + __popcll(m1) - 1);  // left(-64) = 0 always qualifies
                        const int nidx = sidx + 1;
                        // (both candidates read, then a scalar select: written as `sidx < 64 ? readlane(l0) : readlane(l1)` the
                        // compiler branches around each v_readlane - four branches on every symbol's chain)
                        const uint32_t la = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(l0), sidx & 63));
                        const uint32_t lb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(l1), sidx & 63));
                        const uint32_t ra = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(l0), nidx & 63));
                        const uint32_t rb = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(l1), nidx & 63));
                        const uint32_t left = (sidx & 64) ? lb : la;
                        uint32_t right = (nidx & 64) ? rb : ra;
                        if (sidx == kAlphabet - 1) right = 1u << kRcPrecision;
                        rc_dist = uni64(rc_dist - scale * left);
                        rc_range = uni64(scale * static_cast<uint64_t>(right - left));
                        uint32_t range_hi = static_cast<uint32_t>(rc_range >> 32);
                        asm volatile("" : "+s"(range_hi));
                        if (range_hi == 0u) {
                            rc_range <<= 32;
                            rc_dist = (rc_dist << 32) | next_word();
                        }
                        sym_l = lane == i ? sidx : sym_l;
                        ++n_decoded;
                        n_done = i + 1;
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0), "+v"(n1) :: "memory");
                        l0 = n0; l1 = n1;
                    }
                    // (r05: one store per lane here instead of two LDS reads and a one-lane store on every symbol's path; the symbols
                    // decoded before invalid data was met are stored too, as before)
                    if (lane < n_done) lat[s_py[lane] * W + s_px[lane]] = static_cast<int8_t>(sym_l + kAcLo);
                }
                __syncthreads();
                GP_ADD(4);
                if (*s_err != 0) break;
            }
            if (*s_err != 0) break;
        }
        if (*s_err != 0) break;
    }
    if (tid == 0) {
        P.status[0] = *s_err;
        P.status[1] = static_cast<int32_t>(word_pos);
        P.status[2] = static_cast<int32_t>(n_decoded & 0xffffffffu);
        P.status[3] = static_cast<int32_t>(n_decoded >> 32);
#ifdef CCD_GEN_PROFILE
        for (int k = 0; k < 5; ++k) P.status[40 + k] = static_cast<int32_t>(gp[k] >> 10);  // Kticks per phase (thread 0): gather, layers, output, tables, symbols
#endif
    }
}

__global__ __launch_bounds__(kEntThreads) void entropy_kernel(const EntropyParams* slots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const EntropyParams& P = slots[blockIdx.x];
    if (P.narrow) entropy_decode_slot<true>(P, smem_raw);
    else entropy_decode_slot<false>(P, smem_raw);
}

size_t entropy_lds_bytes(int dim, int arm_len) {
    size_t n = static_cast<size_t>((arm_len + 1) & ~1) * 8;
    n += static_cast<size_t>(2 * dim * kChunk + 2 * kChunk) * 8;
    n += static_cast<size_t>(kChunk) * kAlphabet * 4;
    n += static_cast<size_t>(4 * kChunk + 4) * 4;
    n += static_cast<size_t>(kChunk) * 8;  // s_rcpx
    return (n + 15) & ~size_t{15};
}

hipError_t launch_entropy(const EntropyParams* d_slots, int n_slots, size_t lds_bytes, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(entropy_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(entropy_kernel, dim3(n_slots), dim3(kEntThreads), lds_bytes, stream, d_slots);
    return hipGetLastError();
}

// ---- debug: the CDF boundaries exactly as phase B computes them ------------------------------------
__global__ void laplace_bounds_kernel(const int32_t* mu_idx, const int32_t* scale_idx, const int32_t* sym,
                                      const float* scale_table, int64_t n, uint32_t* left, uint32_t* right) {
    // (debug entry points only hold the float32 scale table: RN(1 / b) formed here like the host forms the rcp table)
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = min(max(mu_idx[i], 0), kNumMu - 1), c = min(max(scale_idx[i], 0), kNumScale - 1), s = sym[i];
    const double rcp = 1.0 / static_cast<double>(scale_table[c]);
    left[i] = (s == kAcLo) ? 0u : laplace_left(m, rcp, s);
    right[i] = (s == kAcLo + kAlphabet - 1) ? (1u << kRcPrecision) : laplace_left(m, rcp, s + 1);
}

hipError_t launch_laplace_bounds(const int32_t* mu_idx, const int32_t* scale_idx, const int32_t* sym,
                                 const float* scale_table, int64_t n, uint32_t* left, uint32_t* right, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((n + threads - 1) / threads);
    hipLaunchKernelGGL(laplace_bounds_kernel, dim3(blocks), dim3(threads), 0, stream, mu_idx, scale_idx, sym,
                       scale_table, n, left, right);
    return hipGetLastError();
}

// the same sweep as launch_laplace_sweep_pipe (ccd_entropy_pipe.hip) for this kernel's laplace_left
__global__ void laplace_sweep_generic_kernel(const float* scale_table, int scale_first, int n_scales, uint32_t* out) {
    const int64_t n = static_cast<int64_t>(n_scales) * kNumMu * 127;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int s = static_cast<int>(i % 127) - 63;
        const int64_t r = i / 127;
        out[i] = laplace_left(static_cast<int>(r % kNumMu), 1.0 / static_cast<double>(scale_table[scale_first + static_cast<int>(r / kNumMu)]), s);
    }
}
hipError_t launch_laplace_sweep_generic(const float* scale_table, int scale_first, int n_scales, uint32_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(laplace_sweep_generic_kernel, dim3(256 * 16), dim3(256), 0, stream, scale_table, scale_first, n_scales, out);
    return hipGetLastError();
}

}  // namespace ccd
