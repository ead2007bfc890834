// ccd_png.hip - PNG packing of decoded 8-bit RGB planes on the device.
//
// Reference behaviour: coolchic/io/format/png.py:44-62 (write_png: [1,3,H,W] float in [0,1] -> HWC uint8 -> PIL save,
// i.e. zlib deflate on the host).  The integer planes already exist on the device (ccd_batch_plane); this file turns
// them into the bytes of a .png there, so that only the compressed file crosses PCIe and no host core runs zlib.
// PNG bytes are not normative (any conforming zlib stream of the filtered scanlines is the same picture): the device
// does not imitate zlib's LZ77 choices.  Parity bar (tests/test_gpu_parity.py): the picture PIL reads back is
// pixel-exact, and the bytes equal those of the CPU restatement oracle/png_pack.py (all steps are integer).
//
// Format: signature, IHDR, one IDAT with a zlib stream (0x78 0x01) of the filtered scanlines, IEND.
//   filter   per row the one of None/Sub/Up/Average/Paeth with the smallest sum of absolute signed residuals
//   deflate  rows grouped into blocks of rows_per_block(w) rows (about 32 KB of scanlines); every block is a
//            dynamic-Huffman block of literals only + end-of-block: HLIT = 257, HDIST = 1 (length 0), code-length
//            alphabet = 4-bit codes for the lengths 0..15, no run-length symbols; literal lengths are optimal
//            (Moffat-Katajainen in-place construction on symbols sorted by (count, symbol)), limited to 15 bits
//
// Kernels (all HBM traffic is one read of the planes, one write + one read of the scanlines, one write of the file):
//   png_filter_huff_kernel  one workgroup per deflate block: filter choice, scanlines, per-row Adler sums, histogram in
//                           LDS, bitonic sort of the used symbols, code lengths + canonical codes, bits of the block
//   png_emit_kernel         one workgroup per deflate block: block start = sum of the previous blocks' bits; scanlines
//                           staged in LDS, per-thread bit counts, scan, bit packing (whole words stored, the two
//                           boundary words of a thread merged with atomicOr into the zeroed file)
//   png_trailer_kernel      container bytes, Adler-32 from the row sums
//   png_crc_kernel          CRC-32 of the IDAT chunk as XOR of 512-byte chunk CRCs multiplied by x^(8 * bytes behind)
//   png_crc_final_kernel    stores the CRC, the IEND chunk and the file size
#include <hip/hip_runtime.h>

#include <cstdint>
#include <algorithm>
#include <cstring>
#include <new>

#include "../../include/ccd.h"

namespace {

constexpr int kMaxBits = 15;
constexpr int kHeaderBits = 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4;  // 1106 bits in front of the first literal
constexpr int kBlockTarget = 32768;   // bytes of scanlines per deflate block (at least one row)
constexpr int kCrcChunk = 512;
constexpr int kDataStart = 43;        // signature 8 + IHDR 25 + IDAT length/type 8 + zlib header 2
constexpr int kMaxDim = 16383;        // 14-bit picture sizes (header.py:244-307); one scanline then fits the LDS stage
constexpr int kStageBytes = 49152;    // LDS stage of png_emit_kernel: >= max(kBlockTarget, 3 * kMaxDim + 1)
constexpr uint32_t kPoly = 0xEDB88320u;

inline int rows_per_block(int w) { const int r = kBlockTarget / (3 * w + 1); return r < 1 ? 1 : r; }

constexpr int kMaxBatch = 64;        // pictures per launch set (a longer list is packed in several sets)

struct PngJob {                // one picture (a table of these lives in device memory)
    const uint8_t* plane[3];   // r, g, b: [h][w]
    uint32_t* out;             // the file, 4-byte aligned
    uint64_t cap_bits;
    uint8_t* scan;             // [h][3 w + 1] filtered scanlines
    uint32_t* codes;           // [nblk][257] bit-reversed code | length << 16
    uint32_t* blk_bits;        // [nblk]
    uint32_t* row_adler;       // [h][2] (sum d_i, sum (n - i) d_i) mod 65521
    uint32_t* meta;            // [0] deflate bits, [1] deflate bytes, [2] file bytes, [3] crc accumulator, [4] overflow flag
    int32_t h, w, rows, nblk;  // rows = rows per deflate block
    uint32_t zero_words, pad;  // words of `out` cleared before the bit packer merges into them
};

struct PngBatch {              // kernel argument
    const PngJob* img;
    int32_t n, pad;
    uint32_t blk_prefix[kMaxBatch + 1];  // first workgroup of picture i in the per-block kernels
    uint32_t crc_prefix[kMaxBatch + 1];  // first workgroup of picture i in the CRC kernel
    uint32_t x2n[32];                    // x^(2^k) mod P (CRC-32, reflected)
};

// picture that owns workgroup `wg` (prefix[] sits in the kernel argument segment: scalar loads)
__device__ __forceinline__ int find_image(const uint32_t* prefix, int n, uint32_t wg) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= wg) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int abs_res(int v) { v &= 255; return v < 128 ? v : 256 - v; }

__device__ __forceinline__ int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

__device__ __forceinline__ int filt_one(int f, int x, int a, int b, int c) {
    switch (f) {
        case 0: return x & 255;
        case 1: return (x - a) & 255;
        case 2: return (x - b) & 255;
        case 3: return (x - ((a + b) >> 1)) & 255;
        default: return (x - paeth(a, b, c)) & 255;
    }
}

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// ---- kernel A -----------------------------------------------------------------------------------------------------
constexpr int kThreadsA = 256;   // 1024 threads: single pictures 10-20 % faster, batches and 4K 2x slower (occupancy; measured)

__global__ __launch_bounds__(kThreadsA) void png_filter_huff_kernel(PngBatch B) {
    __shared__ uint32_t s_hist[257];
    __shared__ uint32_t s_key[512];
    __shared__ uint32_t s_len[257];
    __shared__ uint32_t s_A[257];   // scratch of the code-length construction
    __shared__ unsigned long long s_red[kThreadsA / 64][8];
    __shared__ uint32_t s_m, s_sum;
    __shared__ uint32_t s_num[kMaxBits + 1], s_first[kMaxBits + 1], s_base[kMaxBits + 1];
    __shared__ uint32_t s_par[256], s_dep[256], s_cnt[258];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int im = __builtin_amdgcn_readfirstlane(find_image(B.blk_prefix, B.n, blockIdx.x));
    const PngJob J = B.img[im];
    const int k = static_cast<int>(blockIdx.x - B.blk_prefix[im]);
    const int w = J.w, n = 3 * w, N = n + 1;
    {   // this block's share of the file is cleared here (the bit packer of the next kernel merges into zeroed words)
        const uint32_t share = (J.zero_words + J.nblk - 1) / J.nblk;
        const uint32_t z_lo = min(J.zero_words, static_cast<uint32_t>(k) * share), z_hi = min(J.zero_words, z_lo + share);
        for (uint32_t i = z_lo + tid; i < z_hi; i += kThreadsA) J.out[i] = 0;
        if (k == 0 && tid < 8) J.meta[tid] = 0;
    }
    const int y_lo = k * J.rows, y_hi = min(J.h, y_lo + J.rows);
    for (int i = tid; i < 258; i += kThreadsA) { s_cnt[i] = 0; if (i < 257) { s_hist[i] = 0; s_len[i] = 0; } }
    if (tid == 0) { s_m = 0; s_sum = 0; }
    if (tid <= kMaxBits) s_num[tid] = 0;
    __syncthreads();
    for (int y = y_lo; y < y_hi; ++y) {
        // ---- pass 1: cost of the five filters
        unsigned long long cost[5] = {0, 0, 0, 0, 0};
        for (int x = tid; x < w; x += kThreadsA) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint8_t* p = J.plane[c] + static_cast<size_t>(y) * w + x;
                const int cur = p[0];
                const int a = x > 0 ? p[-1] : 0;
                const int b = y > 0 ? p[-w] : 0;
                const int cc = (x > 0 && y > 0) ? p[-w - 1] : 0;
                cost[0] += abs_res(cur);
                cost[1] += abs_res(cur - a);
                cost[2] += abs_res(cur - b);
                cost[3] += abs_res(cur - ((a + b) >> 1));
                cost[4] += abs_res(cur - paeth(a, b, cc));
            }
        }
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            unsigned long long v = cost[f];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0) s_red[wave][f] = v;
        }
        __syncthreads();
        int ftype = 0;
        {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                unsigned long long v = 0;
                for (int q = 0; q < kThreadsA / 64; ++q) v += s_red[q][f];
                if (v < best) { best = v; ftype = f; }
            }
        }
        __syncthreads();
        // ---- pass 2: scanline, histogram, Adler sums
        uint8_t* row = J.scan + static_cast<size_t>(y) * N;
        unsigned long long sa = 0, sb = 0;
        if (tid == 0) {
            row[0] = static_cast<uint8_t>(ftype);
            atomicAdd(&s_hist[ftype], 1u);
            sa += ftype;
            sb += static_cast<unsigned long long>(N) * ftype;
        }
        for (int x = tid; x < w; x += kThreadsA) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint8_t* p = J.plane[c] + static_cast<size_t>(y) * w + x;
                const int cur = p[0];
                const int a = x > 0 ? p[-1] : 0;
                const int b = y > 0 ? p[-w] : 0;
                const int cc = (x > 0 && y > 0) ? p[-w - 1] : 0;
                const int v = filt_one(ftype, cur, a, b, cc);
                const int idx = 1 + 3 * x + c;
                row[idx] = static_cast<uint8_t>(v);
                atomicAdd(&s_hist[v], 1u);
                sa += v;
                sb += static_cast<unsigned long long>(N - idx) * v;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_down(sa, o); sb += __shfl_down(sb, o); }
        if (lane == 0) { s_red[wave][5] = sa; s_red[wave][6] = sb; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long ta = 0, tb = 0;
            for (int q = 0; q < kThreadsA / 64; ++q) { ta += s_red[q][5]; tb += s_red[q][6]; }
            J.row_adler[2 * y] = static_cast<uint32_t>(ta % 65521u);
            J.row_adler[2 * y + 1] = static_cast<uint32_t>(tb % 65521u);
        }
        __syncthreads();
    }
    // ---- used symbols sorted by (count, symbol)
    if (tid == 0) s_hist[256] = 1;
    __syncthreads();
    for (int t = tid; t < 512; t += kThreadsA) {
        uint32_t key = 0xFFFFFFFFu;
        if (t < 257 && s_hist[t] > 0) { key = (s_hist[t] << 9) | static_cast<uint32_t>(t); atomicAdd(&s_m, 1u); }
        s_key[t] = key;
    }
    __syncthreads();
    for (int size = 2; size <= 512; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < 512; t += kThreadsA) {
                const int partner = t ^ stride;
                if (partner > t) {
                    const uint32_t a = s_key[t], b = s_key[partner];
                    const bool up = (t & size) == 0;
                    if ((a > b) == up) { s_key[t] = b; s_key[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
    // ---- Huffman tree over the sorted symbols (one lane: the two-queue merge is sequential; at most 257 symbols).
    // Moffat-Katajainen phase 1, in place on the ascending counts: afterwards s_A[i] is the parent of internal node i.
    const int m = static_cast<int>(s_m);
    if (tid == 0) {
        uint32_t* A = s_A;
        for (int i = 0; i < m; ++i) A[i] = s_key[i] >> 9;
        A[0] += A[1];
        int root = 0, leaf = 2;
        for (int nxt = 1; nxt < m - 1; ++nxt) {
            if (leaf >= m || A[root] < A[leaf]) { A[nxt] = A[root]; A[root++] = nxt; }
            else A[nxt] = A[leaf++];
            if (leaf >= m || (root < nxt && A[root] < A[leaf])) { A[nxt] += A[root]; A[root++] = nxt; }
            else A[nxt] += A[leaf++];
        }
    }
    __syncthreads();
    // ---- depth of the internal nodes 0 .. m-2 (node m-2 is the root) by pointer jumping, one node per thread
    const int ni = m - 1;
    uint32_t par = 0, dep = 0;
    if (tid < ni) {
        par = tid == ni - 1 ? static_cast<uint32_t>(tid) : s_A[tid];
        dep = tid == ni - 1 ? 0u : 1u;
        s_par[tid] = par; s_dep[tid] = dep;
    }
    __syncthreads();
    for (int r = 0; r < 9; ++r) {  // 2^9 > 256 levels
        uint32_t pd = 0, pp = 0;
        if (tid < ni) { pd = s_dep[par]; pp = s_par[par]; }
        __syncthreads();
        if (tid < ni) { dep += pd; par = pp; s_dep[tid] = dep; s_par[tid] = par; }
        __syncthreads();
    }
    // ---- leaves per depth: the two children of every internal node at depth d - 1 are the internal nodes and the leaves at
    // depth d.  Sorted by count the leaves have non-increasing depths, so the counts per length are all that is needed.
    // Limit to 15 bits: longer codes are folded into the limit, then the Kraft excess is worked off.
    if (tid < ni) atomicAdd(&s_cnt[dep], 1u);
    __syncthreads();
    if (tid < 256) {
        const int d = tid + 1;  // 1 .. 256
        const uint32_t leaves = 2u * s_cnt[d - 1] - s_cnt[d];
        if (leaves) atomicAdd(&s_num[min(d, kMaxBits)], leaves);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t total = 0;
        for (int l = 1; l <= kMaxBits; ++l) total += s_num[l] << (kMaxBits - l);
        while (total != (1u << kMaxBits)) {
            s_num[kMaxBits]--;
            for (int l = kMaxBits - 1; l > 0; --l)
                if (s_num[l]) { s_num[l]--; s_num[l + 1] += 2; break; }
            --total;
        }
        // rarest symbols take the longest codes: sorted positions [s_first[l], s_first[l] + num[l]) get length l;
        // canonical codes (RFC 1951 3.2.2): first code of every length
        uint32_t first = 0, code = 0;
        for (int l = kMaxBits; l > 0; --l) { s_first[l] = first; first += s_num[l]; }
        for (int bits = 1; bits <= kMaxBits; ++bits) {
            code = (code + (bits > 1 ? s_num[bits - 1] : 0u)) << 1;
            s_base[bits] = code;
        }
    }
    __syncthreads();
    for (int j = tid; j < m; j += kThreadsA) {
        int l = kMaxBits;
        while (static_cast<uint32_t>(j) >= s_first[l] + s_num[l]) --l;
        s_len[s_key[j] & 511u] = static_cast<uint32_t>(l);
    }
    __syncthreads();
    // ---- canonical codes: within a length, symbols in increasing order
    for (int sy = tid; sy < 257; sy += kThreadsA) {
        const uint32_t l = s_len[sy];
        uint32_t v = 0;
        if (l) {
            uint32_t rank = 0;
            for (int q = 0; q < sy; ++q) rank += s_len[q] == l ? 1u : 0u;
            v = rev_bits(s_base[l] + rank, static_cast<int>(l)) | (l << 16);
        }
        J.codes[static_cast<size_t>(k) * 257 + sy] = v;
        atomicAdd(&s_sum, s_hist[sy] * l);
    }
    __syncthreads();
    if (tid == 0) J.blk_bits[k] = kHeaderBits + s_sum;
}

// ---- kernel B -----------------------------------------------------------------------------------------------------
constexpr int kThreadsB = 256;

__device__ __forceinline__ void or_bits(uint32_t* out, uint64_t pos, uint32_t value, int nbits) {
    if (nbits == 0) return;
    const uint64_t v = static_cast<uint64_t>(value) << (pos & 31);
    atomicOr(&out[pos >> 5], static_cast<uint32_t>(v));
    if ((pos & 31) + nbits > 32) atomicOr(&out[(pos >> 5) + 1], static_cast<uint32_t>(v >> 32));
}

__global__ __launch_bounds__(kThreadsB) void png_emit_kernel(PngBatch B) {
    __shared__ uint8_t s_data[kStageBytes];
    __shared__ uint32_t s_code[257];
    __shared__ unsigned long long s_part[kThreadsB / 64];
    __shared__ uint32_t s_scan[kThreadsB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int im = __builtin_amdgcn_readfirstlane(find_image(B.blk_prefix, B.n, blockIdx.x));
    const PngJob J = B.img[im];
    const int k = static_cast<int>(blockIdx.x - B.blk_prefix[im]);
    const int N = 3 * J.w + 1;
    const int y_lo = k * J.rows, y_hi = min(J.h, y_lo + J.rows);
    const int nb = (y_hi - y_lo) * N;
    // start of the block = bits of all blocks before it
    unsigned long long before = 0;
    for (int i = tid; i < k; i += kThreadsB) before += J.blk_bits[i];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
    if (lane == 0) s_part[wave] = before;
    for (int i = tid; i < 257; i += kThreadsB) s_code[i] = J.codes[static_cast<size_t>(k) * 257 + i];
    {   // scanlines of the block -> LDS (4-byte words where the block start allows it)
        const uint8_t* src = J.scan + static_cast<size_t>(y_lo) * N;
        for (int i = tid; i < nb; i += kThreadsB) s_data[i] = src[i];
    }
    __syncthreads();
    before = 0;
    for (int q = 0; q < kThreadsB / 64; ++q) before += s_part[q];
    const uint32_t my_bits = J.blk_bits[k];
    const uint64_t start = static_cast<uint64_t>(kDataStart) * 8 + before;
    if (k == J.nblk - 1 && tid == 0) J.meta[0] = static_cast<uint32_t>(before + my_bits);
    if (start + my_bits + 256 > J.cap_bits) {  // 20 container bytes follow the last block
        if (tid == 0) J.meta[4] = 1;  // never with a buffer of ccd_png_bound() bytes
        return;
    }
    // ---- block header
    if (tid == 0) {
        const uint32_t final_blk = k == J.nblk - 1 ? 1u : 0u;
        or_bits(J.out, start, final_blk | (2u << 1) | (0u << 3) | (0u << 8) | (15u << 13), 17);
        const uint32_t eob = s_code[256];
        or_bits(J.out, start + my_bits - (eob >> 16), eob & 0xFFFFu, static_cast<int>(eob >> 16));
    }
    if (tid < 19) or_bits(J.out, start + 17 + 3 * tid, tid < 3 ? 0u : 4u, 3);  // lengths of the code-length codes 16, 17, 18, then 0..15
    for (int s = tid; s < 258; s += kThreadsB)
        or_bits(J.out, start + 74 + 4 * s, rev_bits(s < 257 ? (s_code[s] >> 16) : 0u, 4), 4);
    // ---- literals: contiguous chunk per thread
    const int cb = (nb + kThreadsB - 1) / kThreadsB;
    const int lo = min(nb, tid * cb), hi = min(nb, lo + cb);
    uint32_t bits = 0;
    for (int i = lo; i < hi; ++i) bits += s_code[s_data[i]] >> 16;
    // exclusive scan over the workgroup
    uint32_t incl = bits;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    uint32_t offset = incl - bits;
    for (int q = 0; q < wave; ++q) offset += s_scan[q];
    uint64_t pos = start + kHeaderBits + offset;
    uint32_t wi = static_cast<uint32_t>(pos >> 5);
    int nacc = static_cast<int>(pos & 31);
    uint64_t acc = 0;
    bool first = true;
    for (int i = lo; i < hi; ++i) {
        const uint32_t c = s_code[s_data[i]];
        acc |= static_cast<uint64_t>(c & 0xFFFFu) << nacc;
        nacc += static_cast<int>(c >> 16);
        if (nacc >= 32) {
            if (first) atomicOr(&J.out[wi], static_cast<uint32_t>(acc));
            else J.out[wi] = static_cast<uint32_t>(acc);  // every bit of this word belongs to this thread
            first = false;
            ++wi;
            acc >>= 32;
            nacc -= 32;
        }
    }
    if (nacc > 0 && acc != 0) atomicOr(&J.out[wi], static_cast<uint32_t>(acc));
}

// ---- kernel C -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_be32(uint8_t* p, uint32_t v) {
    p[0] = static_cast<uint8_t>(v >> 24); p[1] = static_cast<uint8_t>(v >> 16);
    p[2] = static_cast<uint8_t>(v >> 8); p[3] = static_cast<uint8_t>(v);
}

__global__ __launch_bounds__(256) void png_trailer_kernel(PngBatch B) {
    __shared__ unsigned long long s_r[4][3];
    const PngJob J = B.img[blockIdx.x];
    if (J.meta[4]) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Adler-32 of the scanlines from the row sums.  Row after row it is B += N A + b_r, A += a_r, starting at A = 1, B = 0;
    // in closed form A = 1 + sum a_q and B = sum b_q + N (H + sum a_q (H - 1 - q)), everything modulo 65521.
    unsigned long long sa = 0, sb = 0, sw = 0;
    for (int q = tid; q < J.h; q += 256) {
        const unsigned long long a = J.row_adler[2 * q];
        sa += a;
        sb += J.row_adler[2 * q + 1];
        sw += a * static_cast<unsigned long long>(J.h - 1 - q);  // < 2^16 * 2^14 per term, at most 64 terms per thread
    }
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_down(sa, o); sb += __shfl_down(sb, o); sw += __shfl_down(sw, o); }
    if (lane == 0) { s_r[wave][0] = sa; s_r[wave][1] = sb; s_r[wave][2] = sw; }
    __syncthreads();
    if (tid != 0) return;
    sa = sb = sw = 0;
    for (int q = 0; q < 4; ++q) { sa += s_r[q][0]; sb += s_r[q][1]; sw += s_r[q][2]; }
    const unsigned long long M = 65521ull, N = static_cast<unsigned long long>(3 * J.w + 1) % M;
    const uint32_t ad_a = static_cast<uint32_t>((1 + sa) % M);
    const uint32_t ad_b = static_cast<uint32_t>((sb % M + N * ((static_cast<unsigned long long>(J.h) + sw % M) % M)) % M);
    uint8_t* out = reinterpret_cast<uint8_t*>(J.out);
    const uint32_t n_def = (J.meta[0] + 7u) >> 3;
    J.meta[1] = n_def;
    // signature, IHDR (8-bit RGB, deflate, adaptive filtering, no interlace), IDAT type, zlib header (32 KB window, no
    // preset dictionary, check bits).  Bytes 40..42 share a word with the first deflate byte, which is already there.
    uint8_t head[kDataStart];
    for (int i = 0; i < kDataStart; ++i) head[i] = 0;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    for (int i = 0; i < 8; ++i) head[i] = sig[i];
    head[11] = 13;
    head[12] = 'I'; head[13] = 'H'; head[14] = 'D'; head[15] = 'R';
    put_be32(head + 16, static_cast<uint32_t>(J.w));
    put_be32(head + 20, static_cast<uint32_t>(J.h));
    head[24] = 8; head[25] = 2;
    {
        uint32_t crc = 0xFFFFFFFFu;
        for (int i = 12; i < 29; ++i) {
            crc ^= head[i];
            for (int b = 0; b < 8; ++b) crc = (crc & 1u) ? (crc >> 1) ^ kPoly : crc >> 1;
        }
        put_be32(head + 29, crc ^ 0xFFFFFFFFu);
    }
    head[37] = 'I'; head[38] = 'D'; head[39] = 'A'; head[40] = 'T'; head[41] = 0x78; head[42] = 0x01;
    for (int i = 0; i < 40; ++i) out[i] = head[i];
    for (int i = 40; i < kDataStart; ++i) out[i] |= head[i];
    put_be32(out + 33, 2u + n_def + 4u);
    put_be32(out + kDataStart + n_def, (ad_b << 16) | ad_a);
}

// ---- kernel D -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}

__global__ __launch_bounds__(256) void png_crc_kernel(PngBatch B) {
    // slice-by-4 tables: s_tab[j][v] = CRC of byte v followed by j zero bytes
    __shared__ uint32_t s_tab[4][256];
    {
        uint32_t c = threadIdx.x;
        for (int i = 0; i < 8; ++i) c = (c & 1u) ? (c >> 1) ^ kPoly : c >> 1;
        s_tab[0][threadIdx.x] = c;
    }
    __syncthreads();
    for (int j = 1; j < 4; ++j) {
        const uint32_t c = s_tab[j - 1][threadIdx.x];
        s_tab[j][threadIdx.x] = (c >> 8) ^ s_tab[0][c & 255u];
        __syncthreads();
    }
    const int im = __builtin_amdgcn_readfirstlane(find_image(B.crc_prefix, B.n, blockIdx.x));
    const PngJob J = B.img[im];
    if (J.meta[4]) return;
    // The IDAT chunk's CRC covers its type and data: file bytes [37, end).  Chunks are cut at multiples of kCrcChunk of
    // the FILE offset, so every chunk but the first starts on a word.
    const uint64_t end = 37ull + 4ull + 2ull + J.meta[1] + 4ull;
    const uint64_t c0 = (static_cast<uint64_t>(blockIdx.x - B.crc_prefix[im]) * 256 + threadIdx.x) * kCrcChunk;
    uint32_t part = 0;  // this chunk's term of the file CRC
    if (c0 < end) {
        uint64_t i = c0 < 37 ? 37 : c0;
        const uint64_t hi = min(end, c0 + kCrcChunk);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(J.out);
        uint32_t crc = 0xFFFFFFFFu;
        for (; i < hi && (i & 3); ++i) crc = s_tab[0][(crc ^ p[i]) & 255u] ^ (crc >> 8);
        for (; i + 4 <= hi; i += 4) {
            crc ^= J.out[i >> 2];
            crc = s_tab[3][crc & 255u] ^ s_tab[2][(crc >> 8) & 255u] ^ s_tab[1][(crc >> 16) & 255u] ^ s_tab[0][crc >> 24];
        }
        for (; i < hi; ++i) crc = s_tab[0][(crc ^ p[i]) & 255u] ^ (crc >> 8);
        crc ^= 0xFFFFFFFFu;
        // multiply by x^(8 * bytes behind the chunk)
        uint64_t behind = end - hi;
        uint32_t xp = 1u << 31;
        for (int kbit = 3; behind; behind >>= 1, ++kbit)
            if (behind & 1) xp = multmodp(B.x2n[kbit & 31], xp);
        part = multmodp(xp, crc);
    }
    // one atomic per wave (tens of thousands of chunks otherwise queue up on one address)
    for (int o = 32; o > 0; o >>= 1) part ^= __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0 && part) atomicXor(&J.meta[3], part);
}

__global__ void png_crc_final_kernel(PngBatch B) {
    const PngJob J = B.img[blockIdx.x];
    if (threadIdx.x != 0 || J.meta[4]) return;
    uint8_t* out = reinterpret_cast<uint8_t*>(J.out);
    const uint32_t n_def = J.meta[1];
    uint8_t* p = out + kDataStart + n_def + 4;
    put_be32(p, J.meta[3]);
    const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    for (int i = 0; i < 12; ++i) p[4 + i] = iend[i];
    J.meta[2] = kDataStart + n_def + 4 + 4 + 12;
}

// ---- host side ----------------------------------------------------------------------------------------------------
uint32_t host_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}

template <typename T>
bool grow(T** buf, size_t* cap, size_t need) {  // device buffer of at least `need` elements (contents are scratch)
    if (need <= *cap) return true;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(buf), need * sizeof(T)) != hipSuccess) return false;
    *cap = need;
    return true;
}

}  // namespace

struct ccd_png {
    int device = 0;
    size_t scan_cap = 0, codes_cap = 0, bits_cap = 0, adler_cap = 0, jobs_cap = 0, meta_cap = 0;
    uint8_t* d_scan = nullptr;
    uint32_t* d_codes = nullptr;     // [blocks][257]
    uint32_t* d_blk_bits = nullptr;  // [blocks]
    uint32_t* d_row_adler = nullptr; // [rows][2]
    uint32_t* d_meta = nullptr;      // [pictures][8]
    uint32_t* h_meta = nullptr;      // pinned copy of d_meta
    // job table: two pinned host copies and two device copies used alternately, each guarded by an event recorded behind the
    // kernels that read it - a pack neither blocks on the stream nor hands the runtime a pageable buffer
    PngJob* d_jobs[2] = {nullptr, nullptr};
    PngJob* h_jobs[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ev_used[2] = {false, false};
    int flip = 0;
    uint32_t x2n[32];
    int pending = 0;                 // pictures of the pack in flight
};

extern "C" {

size_t ccd_png_bound(int h, int w) {
    if (h <= 0 || w <= 0 || h > kMaxDim || w > kMaxDim) return 0;
    const size_t raw = static_cast<size_t>(h) * (3 * static_cast<size_t>(w) + 1);
    const size_t nblk = (static_cast<size_t>(h) + rows_per_block(w) - 1) / rows_per_block(w);
    return (raw * 10 + 7) / 8 + nblk * 144 + 128;
}

int ccd_png_create(int device, ccd_png** out) {
    if (!out) return CCD_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return CCD_ERR_HIP;
    ccd_png* p = new (std::nothrow) ccd_png();
    if (!p) return CCD_ERR_NOMEM;
    p->device = device;
    p->x2n[0] = 1u << 30;
    for (int k = 1; k < 32; ++k) p->x2n[k] = host_multmodp(p->x2n[k - 1], p->x2n[k - 1]);
    *out = p;
    return CCD_OK;
}

void ccd_png_destroy(ccd_png* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_scan) (void)hipFree(p->d_scan);
    if (p->d_codes) (void)hipFree(p->d_codes);
    if (p->d_blk_bits) (void)hipFree(p->d_blk_bits);
    if (p->d_row_adler) (void)hipFree(p->d_row_adler);
    if (p->d_meta) (void)hipFree(p->d_meta);
    for (int k = 0; k < 2; ++k) {
        if (p->d_jobs[k]) (void)hipFree(p->d_jobs[k]);
        if (p->h_jobs[k]) (void)hipHostFree(p->h_jobs[k]);
        if (p->ev[k]) (void)hipEventDestroy(p->ev[k]);
    }
    if (p->h_meta) (void)hipHostFree(p->h_meta);
    delete p;
}

int ccd_png_pack_batch(ccd_png* p, const ccd_png_item* items, int n, void* stream) {
    if (!p || !items || n <= 0) return CCD_ERR_ARG;
    size_t scan_need = 0, blocks = 0, rows = 0;
    for (int i = 0; i < n; ++i) {
        const ccd_png_item& it = items[i];
        if (!it.r || !it.g || !it.b || !it.out || it.h <= 0 || it.w <= 0 || it.h > kMaxDim || it.w > kMaxDim) return CCD_ERR_ARG;
        if ((reinterpret_cast<uintptr_t>(it.out) & 3u) || it.cap < ccd_png_bound(it.h, it.w)) return CCD_ERR_ARG;
        const int rpb = rows_per_block(it.w);
        scan_need += (static_cast<size_t>(it.h) * (3 * static_cast<size_t>(it.w) + 1) + 15) & ~static_cast<size_t>(15);
        blocks += (it.h + rpb - 1) / rpb;
        rows += it.h;
    }
    if (hipSetDevice(p->device) != hipSuccess) return CCD_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scan_need > p->scan_cap || blocks * 257 > p->codes_cap || blocks > p->bits_cap || rows * 2 > p->adler_cap ||
        static_cast<size_t>(n) > p->jobs_cap || static_cast<size_t>(n) * 8 > p->meta_cap) {
        if (hipStreamSynchronize(st) != hipSuccess) return CCD_ERR_HIP;  // the previous pack may still use the old workspace
        if (static_cast<size_t>(n) * 8 > p->meta_cap) {
            if (p->h_meta) (void)hipHostFree(p->h_meta);
            p->h_meta = nullptr;
            if (hipHostMalloc(reinterpret_cast<void**>(&p->h_meta), static_cast<size_t>(n) * 8 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
                return CCD_ERR_NOMEM;
        }
        if (!grow(&p->d_scan, &p->scan_cap, scan_need) || !grow(&p->d_codes, &p->codes_cap, blocks * 257) ||
            !grow(&p->d_blk_bits, &p->bits_cap, blocks) || !grow(&p->d_row_adler, &p->adler_cap, rows * 2) ||
            !grow(&p->d_meta, &p->meta_cap, static_cast<size_t>(n) * 8))
            return CCD_ERR_NOMEM;
        if (static_cast<size_t>(n) > p->jobs_cap) {
            for (int k = 0; k < 2; ++k) {
                size_t cap = p->jobs_cap;
                if (!grow(&p->d_jobs[k], &cap, static_cast<size_t>(n))) return CCD_ERR_NOMEM;
                if (p->h_jobs[k]) (void)hipHostFree(p->h_jobs[k]);
                p->h_jobs[k] = nullptr;
                if (hipHostMalloc(reinterpret_cast<void**>(&p->h_jobs[k]), static_cast<size_t>(n) * sizeof(PngJob), hipHostMallocDefault) != hipSuccess)
                    return CCD_ERR_NOMEM;
                if (!p->ev[k] && hipEventCreateWithFlags(&p->ev[k], hipEventDisableTiming) != hipSuccess) return CCD_ERR_HIP;
                p->ev_used[k] = false;  // the stream was drained above
            }
            p->jobs_cap = static_cast<size_t>(n);
        }
    }
    const int jk = p->flip;
    p->flip ^= 1;
    // this copy of the table was last read two packs ago: normally long finished
    if (p->ev_used[jk] && hipEventSynchronize(p->ev[jk]) != hipSuccess) return CCD_ERR_HIP;
    PngJob* jobs = p->h_jobs[jk];
    size_t scan_off = 0, blk_off = 0, row_off = 0;
    for (int i = 0; i < n; ++i) {
        const ccd_png_item& it = items[i];
        PngJob& J = jobs[i];
        std::memset(&J, 0, sizeof(J));
        J.plane[0] = it.r; J.plane[1] = it.g; J.plane[2] = it.b;
        J.out = reinterpret_cast<uint32_t*>(it.out);
        J.cap_bits = static_cast<uint64_t>(it.cap) * 8;
        J.h = it.h; J.w = it.w; J.rows = rows_per_block(it.w); J.nblk = (it.h + J.rows - 1) / J.rows;
        J.scan = p->d_scan + scan_off;
        J.codes = p->d_codes + blk_off * 257;
        J.blk_bits = p->d_blk_bits + blk_off;
        J.row_adler = p->d_row_adler + row_off * 2;
        J.meta = p->d_meta + static_cast<size_t>(i) * 8;
        const size_t bound = ccd_png_bound(it.h, it.w);
        J.zero_words = static_cast<uint32_t>(std::min(it.cap & ~static_cast<size_t>(3), (bound + 3) & ~static_cast<size_t>(3)) / 4);
        scan_off += (static_cast<size_t>(it.h) * (3 * static_cast<size_t>(it.w) + 1) + 15) & ~static_cast<size_t>(15);
        blk_off += J.nblk;
        row_off += it.h;
    }
    hipError_t e = hipMemcpyAsync(p->d_jobs[jk], jobs, static_cast<size_t>(n) * sizeof(PngJob), hipMemcpyHostToDevice, st);
    int rc = e == hipSuccess ? CCD_OK : CCD_ERR_HIP;
    for (int first = 0; first < n && rc == CCD_OK; first += kMaxBatch) {
        const int cnt = std::min(kMaxBatch, n - first);
        PngBatch B;
        std::memset(&B, 0, sizeof(B));
        B.img = p->d_jobs[jk] + first;
        B.n = cnt;
        std::memcpy(B.x2n, p->x2n, sizeof(B.x2n));
        for (int i = 0; i < cnt; ++i) {
            const PngJob& J = jobs[first + i];
            B.blk_prefix[i + 1] = B.blk_prefix[i] + static_cast<uint32_t>(J.nblk);
            const size_t chunks = ccd_png_bound(J.h, J.w) / kCrcChunk + 1;
            B.crc_prefix[i + 1] = B.crc_prefix[i] + static_cast<uint32_t>((chunks + 255) / 256);
        }
        hipLaunchKernelGGL(png_filter_huff_kernel, dim3(B.blk_prefix[cnt]), dim3(kThreadsA), 0, st, B);
        hipLaunchKernelGGL(png_emit_kernel, dim3(B.blk_prefix[cnt]), dim3(kThreadsB), 0, st, B);
        hipLaunchKernelGGL(png_trailer_kernel, dim3(cnt), dim3(256), 0, st, B);
        hipLaunchKernelGGL(png_crc_kernel, dim3(B.crc_prefix[cnt]), dim3(256), 0, st, B);
        hipLaunchKernelGGL(png_crc_final_kernel, dim3(cnt), dim3(64), 0, st, B);
        if (hipGetLastError() != hipSuccess) rc = CCD_ERR_HIP;
    }
    if (hipEventRecord(p->ev[jk], st) != hipSuccess) rc = CCD_ERR_HIP;
    p->ev_used[jk] = true;
    if (rc != CCD_OK) return rc;
    if (hipMemcpyAsync(p->h_meta, p->d_meta, static_cast<size_t>(n) * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
        return CCD_ERR_HIP;
    p->pending = n;
    return CCD_OK;
}

int ccd_png_finish_batch(ccd_png* p, void* stream, int64_t* sizes, int n) {
    if (!p || !sizes || p->pending <= 0 || n != p->pending) return CCD_ERR_ARG;
    if (hipSetDevice(p->device) != hipSuccess) return CCD_ERR_HIP;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return CCD_ERR_HIP;
    p->pending = 0;
    int rc = CCD_OK;
    for (int i = 0; i < n; ++i) {
        const uint32_t* m = p->h_meta + static_cast<size_t>(i) * 8;
        if (m[4]) { sizes[i] = CCD_ERR_NOMEM; rc = CCD_ERR_NOMEM; }  // the file did not fit `cap` (cannot happen with ccd_png_bound())
        else sizes[i] = static_cast<int64_t>(m[2]);
    }
    return rc;
}

int ccd_png_pack(ccd_png* p, const uint8_t* r, const uint8_t* g, const uint8_t* b, int h, int w, uint8_t* out, size_t cap,
                 void* stream) {
    ccd_png_item it;
    it.r = r; it.g = g; it.b = b; it.h = h; it.w = w; it.out = out; it.cap = cap;
    return ccd_png_pack_batch(p, &it, 1, stream);
}

int64_t ccd_png_finish(ccd_png* p, void* stream) {
    int64_t size = 0;
    const int rc = ccd_png_finish_batch(p, stream, &size, 1);
    return rc < 0 ? rc : size;
}

}  // extern "C"
