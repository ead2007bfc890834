// Microbenchmark: cost per symbol of the serial range-decoder chain on ONE wave (MI355X).
// Tables: per symbol 64 left-cumulatives (window) + entry 64 = right bound of the last candidate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int TBL = 80;  // u32 per symbol (65 used, padded)

// Variant A: plain HIP, uniform state in every lane
__global__ __launch_bounds__(64) void chain_a(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    const int lane = threadIdx.x;
    uint32_t wp = 0;
    uint64_t range = ~uint64_t{0};
    uint64_t dist = (uint64_t)words[0] << 32 | words[1];
    wp = 2;
    long long t0 = clock64();
    uint32_t nxt = tbl[lane];
    uint32_t nxt_r = tbl[64];
    for (int i = 0; i < n; ++i) {
        const uint32_t l = nxt;
        const uint32_t top = nxt_r;
        nxt = tbl[(size_t)(i + 1) * TBL + lane];
        nxt_r = tbl[(size_t)(i + 1) * TBL + 64];
        const uint64_t scale = range >> 24;
        const uint64_t prod = scale * l;
        const unsigned long long m = __ballot(prod <= dist);
        const int cnt = __popcll(m);  // >= 1
        const uint64_t pl = __shfl(prod, cnt - 1);
        const uint64_t pr = cnt < 64 ? __shfl(prod, cnt & 63) : scale * top;
        dist -= pl;
        range = pr - pl;
        if ((range >> 32) == 0) { range <<= 32; dist = (dist << 32) | words[wp++]; }
        if (lane == 0) out[i] = (int8_t)cnt;
    }
    long long t1 = clock64();
    if (lane == 0) { stats[0] = (uint64_t)(t1 - t0); stats[1] = wp; stats[2] = dist; }
}

// Variant B: explicit scalarisation with readfirstlane / readlane builtins
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__global__ __launch_bounds__(64) void chain_b(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    const int lane = threadIdx.x;
    uint32_t wp = 2;
    uint32_t r_lo = 0xffffffffu, r_hi = 0xffffffffu;  // range
    uint32_t d_hi = rfl(words[0]), d_lo = rfl(words[1]);
    long long t0 = clock64();
    uint32_t nxt = tbl[lane];
    uint32_t nxt_r = tbl[64];
    for (int i = 0; i < n; ++i) {
        const uint32_t l = nxt;
        const uint32_t top = nxt_r;
        nxt = tbl[(size_t)(i + 1) * TBL + lane];
        nxt_r = tbl[(size_t)(i + 1) * TBL + 64];
        // scale = range >> 24 : 40 bits -> s_lo (32), s_hi (8)
        const uint32_t s_lo = (r_lo >> 24) | (r_hi << 8);
        const uint32_t s_hi = r_hi >> 24;
        // prod = scale * l  (l < 2^25)
        const uint64_t p0 = (uint64_t)s_lo * l;
        const uint32_t p_lo = (uint32_t)p0;
        const uint32_t p_hi = (uint32_t)(p0 >> 32) + s_hi * l;
        const uint64_t prod = ((uint64_t)p_hi << 32) | p_lo;
        const uint64_t dist = ((uint64_t)d_hi << 32) | d_lo;
        const unsigned long long m = __ballot(prod <= dist);
        const int cnt = __popcll(m);
        const uint32_t pl_lo = __builtin_amdgcn_readlane(p_lo, cnt - 1);
        const uint32_t pl_hi = __builtin_amdgcn_readlane(p_hi, cnt - 1);
        uint32_t pr_lo, pr_hi;
        if (cnt < 64) { pr_lo = __builtin_amdgcn_readlane(p_lo, cnt & 63); pr_hi = __builtin_amdgcn_readlane(p_hi, cnt & 63); }
        else { const uint64_t q0 = (uint64_t)s_lo * top; pr_lo = (uint32_t)q0; pr_hi = (uint32_t)(q0 >> 32) + s_hi * top; }
        const uint64_t pl = ((uint64_t)pl_hi << 32) | pl_lo, pr = ((uint64_t)pr_hi << 32) | pr_lo;
        uint64_t nd = dist - pl, nr = pr - pl;
        if ((nr >> 32) == 0) { nr <<= 32; nd = (nd << 32) | rfl(words[wp]); wp++; }
        d_hi = (uint32_t)(nd >> 32); d_lo = (uint32_t)nd; r_hi = (uint32_t)(nr >> 32); r_lo = (uint32_t)nr;
        if (lane == 0) out[i] = (int8_t)cnt;
    }
    long long t1 = clock64();
    if (lane == 0) { stats[0] = (uint64_t)(t1 - t0); stats[1] = wp; stats[2] = ((uint64_t)d_hi << 32) | d_lo; }
}

// Variant C: tables staged in LDS (as in the real kernel), words prefetched into lanes.
constexpr int CH = 128;
__global__ __launch_bounds__(64) void chain_c(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    __shared__ uint32_t s_tbl[CH * 65];
    __shared__ int8_t s_out[CH];
    const int lane = threadIdx.x;
    uint32_t wp = 2;
    uint32_t r_lo = 0xffffffffu, r_hi = 0xffffffffu;
    uint32_t d_hi = rfl(words[0]), d_lo = rfl(words[1]);
    uint32_t wbuf = words[2 + lane];  // 64 words ahead, one per lane
    uint32_t wbase = 2;
    long long total = 0;
    for (int c0 = 0; c0 < n; c0 += CH) {
        for (int j = lane; j < CH * 65; j += 64) s_tbl[j] = tbl[(size_t)(c0 + j / 65) * TBL + j % 65];
        __syncthreads();
        long long t0 = clock64();
        uint32_t l = s_tbl[lane];
        uint32_t top = s_tbl[64];
        for (int i = 0; i < CH; ++i) {
            const uint32_t ln = s_tbl[(i + 1 < CH ? i + 1 : i) * 65 + lane];
            const uint32_t topn = s_tbl[(i + 1 < CH ? i + 1 : i) * 65 + 64];
            const uint32_t s_lo = (r_lo >> 24) | (r_hi << 8);
            const uint32_t s_hi = r_hi >> 24;
            const uint64_t p0 = (uint64_t)s_lo * l;
            const uint32_t p_lo = (uint32_t)p0;
            const uint32_t p_hi = (uint32_t)(p0 >> 32) + s_hi * l;
            const uint64_t prod = ((uint64_t)p_hi << 32) | p_lo;
            const uint64_t dist = ((uint64_t)d_hi << 32) | d_lo;
            const unsigned long long m = __ballot(prod <= dist);
            const int cnt = __popcll(m);
            const uint32_t pl_lo = __builtin_amdgcn_readlane(p_lo, cnt - 1);
            const uint32_t pl_hi = __builtin_amdgcn_readlane(p_hi, cnt - 1);
            uint32_t pr_lo, pr_hi;
            if (cnt < 64) { pr_lo = __builtin_amdgcn_readlane(p_lo, cnt & 63); pr_hi = __builtin_amdgcn_readlane(p_hi, cnt & 63); }
            else { const uint64_t q0 = (uint64_t)s_lo * top; pr_lo = (uint32_t)q0; pr_hi = (uint32_t)(q0 >> 32) + s_hi * top; }
            const uint64_t pl = ((uint64_t)pl_hi << 32) | pl_lo, pr = ((uint64_t)pr_hi << 32) | pr_lo;
            uint64_t nd = dist - pl, nr = pr - pl;
            if ((nr >> 32) == 0) {
                nr <<= 32;
                nd = (nd << 32) | (uint32_t)__builtin_amdgcn_readlane(wbuf, (wp - wbase) & 63);
                wp++;
                if (wp - wbase == 64) { wbase = wp; wbuf = words[wp + lane]; }
            }
            d_hi = (uint32_t)(nd >> 32); d_lo = (uint32_t)nd; r_hi = (uint32_t)(nr >> 32); r_lo = (uint32_t)nr;
            if (lane == 0) s_out[i] = (int8_t)cnt;
            l = ln; top = topn;
        }
        total += clock64() - t0;
        __syncthreads();
        for (int j = lane; j < CH; j += 64) out[c0 + j] = s_out[j];
    }
    if (lane == 0) { stats[0] = (uint64_t)total; stats[1] = wp; stats[2] = ((uint64_t)d_hi << 32) | d_lo; }
}

// Variant D: L and P (=R-L) tables, no top special case, symbols accumulated with v_writelane, unroll 4.
__global__ __launch_bounds__(64) void chain_d(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    __shared__ uint32_t s_l[CH * 64];
    __shared__ uint32_t s_p[CH * 64];
    __shared__ int8_t s_out[CH];
    const int lane = threadIdx.x;
    uint32_t wp = 2;
    uint64_t range = ~uint64_t{0};
    uint64_t dist = ((uint64_t)rfl(words[0]) << 32) | rfl(words[1]);
    uint32_t wbuf = words[2 + lane];
    uint32_t wbase = 2;
    long long total = 0;
    for (int c0 = 0; c0 < n; c0 += CH) {
        for (int j = lane; j < CH * 64; j += 64) {
            const uint32_t a = tbl[(size_t)(c0 + j / 64) * TBL + j % 64], b = tbl[(size_t)(c0 + j / 64) * TBL + j % 64 + 1];
            s_l[j] = a; s_p[j] = b - a;
        }
        __syncthreads();
        long long t0 = clock64();
        for (int i0 = 0; i0 < CH; i0 += 64) {
            int acc = 0;
            uint32_t l1 = s_l[(i0 + 0) * 64 + lane], p1 = s_p[(i0 + 0) * 64 + lane];
            uint32_t l2 = s_l[(i0 + 1) * 64 + lane], p2 = s_p[(i0 + 1) * 64 + lane];
#pragma unroll 4
            for (int i = 0; i < 64; ++i) {
                const uint32_t l = l1, p = p1;
                l1 = l2; p1 = p2;
                { const int nx = (i0 + i + 2 < CH) ? (i0 + i + 2) : (CH - 1); l2 = s_l[nx * 64 + lane]; p2 = s_p[nx * 64 + lane]; }
                const uint32_t s_lo = (uint32_t)(range >> 24);
                const uint32_t s_hi = (uint32_t)(range >> 56);
                const uint64_t p0 = (uint64_t)s_lo * l;
                const uint32_t p_lo = (uint32_t)p0;
                const uint32_t p_hi = (uint32_t)(p0 >> 32) + s_hi * l;
                const uint64_t prod = ((uint64_t)p_hi << 32) | p_lo;
                const unsigned long long m = __ballot(prod <= dist);
                const int idx = __popcll(m) - 1;
                const uint32_t pl_lo = (uint32_t)__builtin_amdgcn_readlane(p_lo, idx);
                const uint32_t pl_hi = (uint32_t)__builtin_amdgcn_readlane(p_hi, idx);
                const uint32_t psel = (uint32_t)__builtin_amdgcn_readlane(p, idx);
                dist -= ((uint64_t)pl_hi << 32) | pl_lo;
                range = (uint64_t)s_lo * psel + ((uint64_t)(s_hi * psel) << 32);
                if (__builtin_expect((range >> 32) == 0, 0)) {
                    range <<= 32;
                    dist = (dist << 32) | (uint32_t)__builtin_amdgcn_readlane(wbuf, (wp - wbase) & 63);
                    wp++;
                    if (wp - wbase == 64) { wbase = wp; wbuf = words[wp + lane]; }
                }
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(acc) : "s"(idx + 1), "s"(i));
            }
            s_out[i0 + lane] = (int8_t)acc;
        }
        total += clock64() - t0;
        __syncthreads();
        for (int j = lane; j < CH; j += 64) out[c0 + j] = s_out[j];
    }
    if (lane == 0) { stats[0] = (uint64_t)total; stats[1] = wp; stats[2] = dist; }
}

// Variant E: the production hand-written loop (ccd_entropy_pipe.hip) on (L,P) pairs, batches of 16 symbols.
__global__ __launch_bounds__(64) void chain_e(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    __shared__ uint2 s_lp[CH * 64];
    __shared__ int8_t s_out[CH];
    const int lane = threadIdx.x;
    uint32_t wp = 2;
    uint64_t rc_range = ~uint64_t{0};
    uint64_t rc_dist = ((uint64_t)rfl(words[0]) << 32) | rfl(words[1]);
    uint32_t wbuf = words[2 + lane];
    uint32_t wbase = 2;
    long long total = 0;
    for (int c0 = 0; c0 < n; c0 += CH) {
        for (int j = lane; j < CH * 64; j += 64) {
            // descending order in lanes: lane k holds candidate 63 - k
            const int sym = j / 64, k = j % 64, cand = 63 - k;
            const uint32_t a = tbl[(size_t)(c0 + sym) * TBL + cand], b = tbl[(size_t)(c0 + sym) * TBL + cand + 1];
            s_lp[j] = make_uint2(a, b - a);
        }
        __syncthreads();
        long long t0 = clock64();
        for (int b0 = 0; b0 < CH; b0 += 16) {
            int raw = 0;
            uint32_t i = 0;
            const uint32_t cnt = 16;
            const uint32_t tab_addr = (uint32_t)(uintptr_t)(s_lp + b0 * 64 + lane);
            while (i < cnt) {
                uint32_t status, k_rare;
                uint32_t taddr = tab_addr + i * 512u;
                asm volatile(
                    "s_mov_b64 s[50:51], %[dst]\n\t"
                    "s_mov_b64 s[52:53], %[rng]\n\t"
                    "ds_read_b64 v[40:41], %[ta]\n\t"
                    "1:\n\t"
                    "ds_read_b64 v[42:43], %[ta] offset:512\n\t"
                    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_mad_u64_u32 v[44:45], s[42:43], s40, v40, 0\n\t"
                    "v_mad_u32_u24 v45, v40, s41, v45\n\t"
                    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"
                    "s_ff1_i32_b64 s44, vcc\n\t"
                    "v_readlane_b32 s45, v41, s44\n\t"
                    "v_readlane_b32 s46, v44, s44\n\t"
                    "v_readlane_b32 s47, v45, s44\n\t"
                    "s_mul_i32 s48, s40, s45\n\t"
                    "s_mul_hi_u32 s49, s40, s45\n\t"
                    "s_mul_i32 s45, s41, s45\n\t"
                    "s_add_u32 s49, s49, s45\n\t"
                    "s_cmp_eq_u32 s49, 0\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_sub_u32 s50, s50, s46\n\t"
                    "s_subb_u32 s51, s51, s47\n\t"
                    "s_mov_b64 s[52:53], s[48:49]\n\t"
                    "s_mov_b32 m0, %[i]\n\t"
                    "v_writelane_b32 %[raw], s44, m0\n\t"
                    "s_add_u32 %[i], %[i], 1\n\t"
                    "s_cmp_lt_u32 %[i], %[cnt]\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "ds_read_b64 v[40:41], %[ta] offset:1024\n\t"
                    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_mad_u64_u32 v[44:45], s[42:43], s40, v42, 0\n\t"
                    "v_mad_u32_u24 v45, v42, s41, v45\n\t"
                    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"
                    "s_ff1_i32_b64 s44, vcc\n\t"
                    "v_readlane_b32 s45, v43, s44\n\t"
                    "v_readlane_b32 s46, v44, s44\n\t"
                    "v_readlane_b32 s47, v45, s44\n\t"
                    "s_mul_i32 s48, s40, s45\n\t"
                    "s_mul_hi_u32 s49, s40, s45\n\t"
                    "s_mul_i32 s45, s41, s45\n\t"
                    "s_add_u32 s49, s49, s45\n\t"
                    "s_cmp_eq_u32 s49, 0\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_sub_u32 s50, s50, s46\n\t"
                    "s_subb_u32 s51, s51, s47\n\t"
                    "s_mov_b64 s[52:53], s[48:49]\n\t"
                    "s_mov_b32 m0, %[i]\n\t"
                    "v_writelane_b32 %[raw], s44, m0\n\t"
                    "s_add_u32 %[i], %[i], 1\n\t"
                    "v_add_u32 %[ta], 0x400, %[ta]\n\t"
                    "s_cmp_lt_u32 %[i], %[cnt]\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "2:\n\t"
                    "s_mov_b32 %[st], 0\n\t"
                    "s_branch 4f\n\t"
                    "3:\n\t"
                    "s_mov_b32 %[st], 1\n\t"
                    "4:\n\t"
                    "s_mov_b32 %[kr], s44\n\t"
                    "s_mov_b64 %[dst], s[50:51]\n\t"
                    "s_mov_b64 %[rng], s[52:53]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [i] "+s"(i), [raw] "+v"(raw), [ta] "+v"(taddr), [st] "=s"(status),
                      [kr] "=s"(k_rare)
                    : [cnt] "s"(cnt)
                    : "memory", "vcc", "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
                      "v40", "v41", "v42", "v43", "v44", "v45");
                if (status == 0) break;
                const uint2 cur = s_lp[(b0 + i) * 64 + lane];
                const uint32_t sc_lo = (uint32_t)(rc_range >> 24), sc_hi = (uint32_t)(rc_range >> 56);
                const uint64_t p0 = (uint64_t)sc_lo * cur.x;
                const uint32_t p_lo = (uint32_t)p0, p_hi = (uint32_t)(p0 >> 32) + sc_hi * cur.x;
                const int k = (int)k_rare;
                const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(p_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(p_lo, k);
                const uint32_t psel = (uint32_t)__builtin_amdgcn_readlane(cur.y, k);
                uint64_t nd = rc_dist - pl;
                uint64_t nr = (uint64_t)sc_lo * psel + ((uint64_t)(sc_hi * psel) << 32);
                nr <<= 32;
                nd = (nd << 32) | (uint32_t)__builtin_amdgcn_readlane(wbuf, (wp - wbase) & 63);
                wp++;
                if (wp - wbase == 64) { wbase = wp; wbuf = words[wp + lane]; }
                rc_dist = ((uint64_t)rfl((uint32_t)(nd >> 32)) << 32) | rfl((uint32_t)nd);
                rc_range = ((uint64_t)rfl((uint32_t)(nr >> 32)) << 32) | rfl((uint32_t)nr);
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(raw) : "s"(k), "s"(i));
                ++i;
            }
            if (lane < 16) s_out[b0 + lane] = (int8_t)(64 - raw);
        }
        total += clock64() - t0;
        __syncthreads();
        for (int j = lane; j < CH; j += 64) out[c0 + j] = s_out[j];
    }
    if (lane == 0) { stats[0] = (uint64_t)total; stats[1] = wp; stats[2] = rc_dist; }
}

// Variant F: trimmed loop (prodP in VALU, m0 as counter, alternating range registers) (ccd_entropy_pipe.hip) on (L,P) pairs, batches of 16 symbols.
__global__ __launch_bounds__(64) void chain_f(const uint32_t* __restrict__ tbl, const uint32_t* __restrict__ words, int n,
                                             int8_t* out, uint64_t* stats) {
    __shared__ uint2 s_lp[CH * 64];
    __shared__ int8_t s_out[CH];
    const int lane = threadIdx.x;
    uint32_t wp = 2;
    uint64_t rc_range = ~uint64_t{0};
    uint64_t rc_dist = ((uint64_t)rfl(words[0]) << 32) | rfl(words[1]);
    uint32_t wbuf = words[2 + lane];
    uint32_t wbase = 2;
    long long total = 0;
    for (int c0 = 0; c0 < n; c0 += CH) {
        for (int j = lane; j < CH * 64; j += 64) {
            // descending order in lanes: lane k holds candidate 63 - k
            const int sym = j / 64, k = j % 64, cand = 63 - k;
            const uint32_t a = tbl[(size_t)(c0 + sym) * TBL + cand], b = tbl[(size_t)(c0 + sym) * TBL + cand + 1];
            s_lp[j] = make_uint2(a, b - a);
        }
        __syncthreads();
        long long t0 = clock64();
        for (int b0 = 0; b0 < CH; b0 += 16) {
            int raw = 0;
            uint32_t i = 0;
            const uint32_t cnt = 16;
            const uint32_t tab_addr = (uint32_t)(uintptr_t)(s_lp + b0 * 64 + lane);
            while (i < cnt) {
                uint32_t status, k_rare;
                uint32_t taddr = tab_addr + i * 512u;
                asm volatile(
                    "s_mov_b64 s[50:51], %[dst]\n\t"
                    "s_mov_b64 s[52:53], %[rng]\n\t"
                    "s_mov_b32 m0, %[i]\n\t"
                    "ds_read_b64 v[40:41], %[ta]\n\t"
                    "1:\n\t"
                    "ds_read_b64 v[42:43], %[ta] offset:512\n\t"
                    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_mad_u64_u32 v[44:45], s[42:43], s40, v40, 0\n\t"
                    "v_mad_u64_u32 v[46:47], s[42:43], s40, v41, 0\n\t"
                    "v_mad_u32_u24 v45, v40, s41, v45\n\t"
                    "v_mad_u32_u24 v47, v41, s41, v47\n\t"
                    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"
                    "s_ff1_i32_b64 s44, vcc\n\t"
                    "v_readlane_b32 s49, v47, s44\n\t"
                    "v_readlane_b32 s48, v46, s44\n\t"
                    "v_readlane_b32 s46, v44, s44\n\t"
                    "v_readlane_b32 s47, v45, s44\n\t"
                    "v_writelane_b32 %[raw], s44, m0\n\t"
                    "s_add_u32 m0, m0, 1\n\t"
                    "s_cmp_eq_u32 s49, 0\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_sub_u32 s50, s50, s46\n\t"
                    "s_subb_u32 s51, s51, s47\n\t"
                    "s_mov_b64 s[52:53], s[48:49]\n\t"
                    "s_cmp_lt_u32 m0, %[cnt]\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "ds_read_b64 v[40:41], %[ta] offset:1024\n\t"
                    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_mad_u64_u32 v[44:45], s[42:43], s40, v42, 0\n\t"
                    "v_mad_u64_u32 v[46:47], s[42:43], s40, v43, 0\n\t"
                    "v_mad_u32_u24 v45, v42, s41, v45\n\t"
                    "v_mad_u32_u24 v47, v43, s41, v47\n\t"
                    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"
                    "s_ff1_i32_b64 s44, vcc\n\t"
                    "v_readlane_b32 s49, v47, s44\n\t"
                    "v_readlane_b32 s48, v46, s44\n\t"
                    "v_readlane_b32 s46, v44, s44\n\t"
                    "v_readlane_b32 s47, v45, s44\n\t"
                    "v_writelane_b32 %[raw], s44, m0\n\t"
                    "s_add_u32 m0, m0, 1\n\t"
                    "s_cmp_eq_u32 s49, 0\n\t"
                    "s_cbranch_scc1 3f\n\t"
                    "s_sub_u32 s50, s50, s46\n\t"
                    "s_subb_u32 s51, s51, s47\n\t"
                    "s_mov_b64 s[52:53], s[48:49]\n\t"
                    "v_add_u32 %[ta], 0x400, %[ta]\n\t"
                    "s_cmp_lt_u32 m0, %[cnt]\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "2:\n\t"
                    "s_mov_b32 %[st], 0\n\t"
                    "s_mov_b32 %[i], m0\n\t"
                    "s_branch 4f\n\t"
                    "3:\n\t"
                    "s_mov_b32 %[st], 1\n\t"
                    "s_sub_u32 %[i], m0, 1\n\t"
                    "4:\n\t"
                    "s_mov_b32 %[kr], s44\n\t"
                    "s_mov_b64 %[dst], s[50:51]\n\t"
                    "s_mov_b64 %[rng], s[52:53]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [i] "+s"(i), [raw] "+v"(raw), [ta] "+v"(taddr), [st] "=s"(status),
                      [kr] "=s"(k_rare)
                    : [cnt] "s"(cnt)
                    : "memory", "vcc", "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
                      "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
                if (status == 0) break;
                const uint2 cur = s_lp[(b0 + i) * 64 + lane];
                const uint32_t sc_lo = (uint32_t)(rc_range >> 24), sc_hi = (uint32_t)(rc_range >> 56);
                const uint64_t p0 = (uint64_t)sc_lo * cur.x;
                const uint32_t p_lo = (uint32_t)p0, p_hi = (uint32_t)(p0 >> 32) + sc_hi * cur.x;
                const int k = (int)k_rare;
                const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(p_hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(p_lo, k);
                const uint32_t psel = (uint32_t)__builtin_amdgcn_readlane(cur.y, k);
                uint64_t nd = rc_dist - pl;
                uint64_t nr = (uint64_t)sc_lo * psel + ((uint64_t)(sc_hi * psel) << 32);
                nr <<= 32;
                nd = (nd << 32) | (uint32_t)__builtin_amdgcn_readlane(wbuf, (wp - wbase) & 63);
                wp++;
                if (wp - wbase == 64) { wbase = wp; wbuf = words[wp + lane]; }
                rc_dist = ((uint64_t)rfl((uint32_t)(nd >> 32)) << 32) | rfl((uint32_t)nd);
                rc_range = ((uint64_t)rfl((uint32_t)(nr >> 32)) << 32) | rfl((uint32_t)nr);
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(raw) : "s"(k), "s"(i));
                ++i;
            }
            if (lane < 16) s_out[b0 + lane] = (int8_t)(64 - raw);
        }
        total += clock64() - t0;
        __syncthreads();
        for (int j = lane; j < CH; j += 64) out[c0 + j] = s_out[j];
    }
    if (lane == 0) { stats[0] = (uint64_t)total; stats[1] = wp; stats[2] = rc_dist; }
}

int main() {
    const int n = 200064;
    std::vector<uint32_t> tbl((size_t)(n + 2) * TBL, 0);
    srand(1);
    for (int i = 0; i < n + 1; ++i) {
        // quantised Laplace window: mu in (31,33), scale log-uniform 0.05..3
        double mu = 31.5 + (rand() % 1000) / 1000.0, b = 0.05 * pow(60.0, (rand() % 1000) / 1000.0);
        for (int j = 0; j <= 64; ++j) {
            double x = j - 0.5;
            double cdf = x <= mu ? 0.5 * exp((x - mu) / b) : 1.0 - 0.5 * exp((mu - x) / b);
            uint32_t v = (uint32_t)(16777088.0 * cdf) + j;
            if (j == 0) v = 0;
            if (j == 64) v = 1u << 24;
            tbl[(size_t)i * TBL + j] = v;
        }
    }
    std::vector<uint32_t> words(n / 2 + 1024);
    for (auto& w : words) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t *d_tbl, *d_words; int8_t* d_out; uint64_t* d_stats;
    CHECK(hipMalloc(&d_tbl, tbl.size() * 4)); CHECK(hipMalloc(&d_words, words.size() * 4));
    CHECK(hipMalloc(&d_out, n)); CHECK(hipMalloc(&d_stats, 64));
    CHECK(hipMemcpy(d_tbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice));
    uint64_t st[3];
    std::vector<int8_t> oa(n), ob(n);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain_a, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(oa.data(), d_out, n, hipMemcpyDeviceToHost));
        printf("chain_a: %.1f clk/symbol (clock64 ticks), words %llu (%.3f bits/sym), dist %llx\n", (double)st[0] / n,
               (unsigned long long)st[1], st[1] * 32.0 / n, (unsigned long long)st[2]);
        hipLaunchKernelGGL(chain_b, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_out, n, hipMemcpyDeviceToHost));
        printf("chain_b: %.1f clk/symbol, words %llu, dist %llx, same symbols: %d\n", (double)st[0] / n, (unsigned long long)st[1],
               (unsigned long long)st[2], (int)(oa == ob));
    }
    hipLaunchKernelGGL(chain_c, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_out, n, hipMemcpyDeviceToHost));
    printf("chain_c (LDS tables): %.1f clk/symbol, words %llu, dist %llx, same symbols: %d\n", (double)st[0] / n, (unsigned long long)st[1],
           (unsigned long long)st[2], (int)(oa == ob));
    hipLaunchKernelGGL(chain_d, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_out, n, hipMemcpyDeviceToHost));
    printf("chain_d (L+P tables, writelane, unroll4): %.1f clk/symbol, words %llu, dist %llx, same symbols: %d\n", (double)st[0] / n, (unsigned long long)st[1],
           (unsigned long long)st[2], (int)(oa == ob));
    hipLaunchKernelGGL(chain_e, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_out, n, hipMemcpyDeviceToHost));
    printf("chain_e (production asm loop): %.1f clk/symbol, words %llu, dist %llx, same symbols: %d\n", (double)st[0] / n, (unsigned long long)st[1],
           (unsigned long long)st[2], (int)(oa == ob));
    hipLaunchKernelGGL(chain_f, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(st, d_stats, 24, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), d_out, n, hipMemcpyDeviceToHost));
    printf("chain_f (trimmed asm loop): %.1f clk/symbol, words %llu, dist %llx, same symbols: %d\n", (double)st[0] / n, (unsigned long long)st[1],
           (unsigned long long)st[2], (int)(oa == ob));
    // wall time for reference
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_b, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain_b wall: %.3f ms -> %.1f ns/symbol\n", ms, ms * 1e6 / n);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_a, dim3(1), dim3(64), 0, 0, d_tbl, d_words, n, d_out, d_stats);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("chain_a wall: %.3f ms -> %.1f ns/symbol\n", ms, ms * 1e6 / n);
    return 0;
}
