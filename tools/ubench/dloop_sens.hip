// Speculation v2 for the decoder's symbol loop: per symbol, test the MODE's window lane (lane 7 of a
// narrow window = round(mu)) on the SCALAR unit - scale * L and scale * P as s_mul_i32 / s_mul_hi_u32, two 64-bit
// subtractions, no VALU <-> SALU hand-over on the chain - and fall through on a hit; the first miss of a batch continues in
// the production search (v_mad / v_cmp / s_ff1 / v_readlane) for the rest of the batch.  The (L, P) of the next row's lane 7
// is read (v_readlane) one symbol ahead.  Measures ticks / symbol of a 16-symbol batch for a given number of leading hits.
//     hipcc --offload-arch=gfx950 -O3 -o dloop_spec dloop_spec.hip && ./dloop_spec
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// current row of symbol j lives in pair j % 3 of (v40:41, v42:43, v46:47); row j + 2 is loaded into pair (j + 2) % 3
#define RL0 "v40"
#define RP0 "v41"
#define RL1 "v42"
#define RP1 "v43"
#define RL2 "v46"
#define RP2 "v47"
#define PAIR0 "v[40:41]"
#define PAIR1 "v[42:43]"
#define PAIR2 "v[46:47]"

#ifdef CMPX
#define SEARCHTAIL(CP)                                                \
    "v_cmpx_ge_u64 vcc, s[50:51], v[44:45]\n\t"                       \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CP ", 0\n\t"            \
    "v_mad_u32_u24 v49, " CP ", s41, v49\n\t"                         \
    "s_ff1_i32_b64 s44, vcc\n\t"                                      \
    "s_nop 0\n\t"                                                     \
    "v_readfirstlane_b32 s48, v48\n\t"                                \
    "v_readfirstlane_b32 s49, v49\n\t"                                \
    "v_readfirstlane_b32 s46, v44\n\t"                                \
    "v_readfirstlane_b32 s47, v45\n\t"                                \
    "s_mov_b64 exec, -1\n\t"                                          \

#else
#define SEARCHTAIL(CP)                                                \
    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"                        \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CP ", 0\n\t"            \
    "v_mad_u32_u24 v49, " CP ", s41, v49\n\t"                         \
    "s_ff1_i32_b64 s44, vcc\n\t"                                      \
    "v_readlane_b32 s48, v48, s44\n\t"                                \
    "v_readlane_b32 s49, v49, s44\n\t"                                \
    "v_readlane_b32 s46, v44, s44\n\t"                                \
    "v_readlane_b32 s47, v45, s44\n\t"                                \

#endif
// production body of symbol J (row in CL / CP, prefetch of row J + 2 into NXT at OFF): entry VAL##J is behind the prefetch
#ifdef ALT
// RIN = pair holding the current range, ROUT = pair that receives the new one (roles swap every symbol)
#ifdef LANEID
#define BOOK(J) "s_nop 1\n\t"
#define BOOK2(J) "v_readfirstlane_b32 s44, %[lid]\n\t"
#define BOOK3(J) "v_writelane_b32 %[raw], s44, " #J "\n\t"
#elif defined(NOBOOK)
#define BOOK2(J)
#define BOOK3(J)
#define BOOK(J) "s_nop 1\n\t"
#else
#define BOOK2(J)
#define BOOK3(J)
#define BOOK(J) "s_ff1_i32_b64 s44, vcc\n\tv_writelane_b32 %[raw], s44, " #J "\n\t"
#endif
#ifdef NOBR
#define RARE(H)
#else
#define RARE(H) "s_cmp_eq_u32 " H ", 0\n\ts_cbranch_scc1 rare_%=\n\t"
#endif
#ifdef NOHI
#define NOPFILL "s_nop 0\n\t"
#define HI_L(CL)
#define HI_P(CP)
#else
#define NOPFILL
#define HI_L(CL) "v_mad_u32_u24 v45, " CL ", s41, v45\n\t"
#define HI_P(CP) "v_mad_u32_u24 v49, " CP ", s41, v49\n\t"
#endif
#ifdef PRELOAD
#define FETCH(NXT, OFF)
#define ROWL(J, CL) "v" STR2(64 + 2 * J)
#define ROWP(J, CP) "v" STR2(65 + 2 * J)
#else
#ifdef NOFETCH
#define FETCH(NXT, OFF)
#else
#define FETCH(NXT, OFF) "ds_read_b64 " NXT ", v50 offset:" OFF "\n\ts_waitcnt lgkmcnt(2)\n\t"
#endif
#define ROWL(J, CL) CL
#define ROWP(J, CP) CP
#endif
#define VBODY2(J, CL, CP, NXT, OFF, RIN, RINHI, ROUT_LO, ROUT_HI)     \
    "s_lshr_b64 s[40:41], " RIN ", 24\n\t"                            \
    FETCH(NXT, OFF)                                                   \
    "val" #J "_%=:\n\t"                                               \
    "v_mad_u64_u32 v[44:45], s[42:43], s40, " CL ", 0\n\t"            \
    HI_L(CL)                                                          \
    "v_cmpx_ge_u64 vcc, s[50:51], v[44:45]\n\t"                       \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CP ", 0\n\t"            \
    HI_P(CP)                                                          \
    BOOK(J)                                                           \
    NOPFILL                                                           \
    "v_readfirstlane_b32 " ROUT_LO ", v48\n\t"                        \
    "v_readfirstlane_b32 " ROUT_HI ", v49\n\t"                        \
    "v_readfirstlane_b32 s46, v44\n\t"                                \
    "v_readfirstlane_b32 s47, v45\n\t"                                \
    BOOK2(J)                                                          \
    "s_mov_b64 exec, -1\n\t"                                          \
    RARE(ROUT_HI)                                                     \
    "s_sub_u32 s50, s50, s46\n\t"                                     \
    "s_subb_u32 s51, s51, s47\n\t"                                    \
    BOOK3(J)
#define VBODY(J, CL, CP, NXT, OFF) VBODY_##J(CL, CP, NXT, OFF)
#define VB_E(J, CL, CP, NXT, OFF) VBODY2(J, CL, CP, NXT, OFF, "s[52:53]", "s53", "s54", "s55")
#define VB_O(J, CL, CP, NXT, OFF) VBODY2(J, CL, CP, NXT, OFF, "s[54:55]", "s55", "s52", "s53")
#define VBODY_0(a,b,c,d) VB_E(0,a,b,c,d)
#define VBODY_1(a,b,c,d) VB_O(1,a,b,c,d)
#define VBODY_2(a,b,c,d) VB_E(2,a,b,c,d)
#define VBODY_3(a,b,c,d) VB_O(3,a,b,c,d)
#define VBODY_4(a,b,c,d) VB_E(4,a,b,c,d)
#define VBODY_5(a,b,c,d) VB_O(5,a,b,c,d)
#define VBODY_6(a,b,c,d) VB_E(6,a,b,c,d)
#define VBODY_7(a,b,c,d) VB_O(7,a,b,c,d)
#define VBODY_8(a,b,c,d) VB_E(8,a,b,c,d)
#define VBODY_9(a,b,c,d) VB_O(9,a,b,c,d)
#define VBODY_10(a,b,c,d) VB_E(10,a,b,c,d)
#define VBODY_11(a,b,c,d) VB_O(11,a,b,c,d)
#define VBODY_12(a,b,c,d) VB_E(12,a,b,c,d)
#define VBODY_13(a,b,c,d) VB_O(13,a,b,c,d)
#define VBODY_14(a,b,c,d) VB_E(14,a,b,c,d)
#define VBODY_15(a,b,c,d) VB_O(15,a,b,c,d)
#else
#define VBODY(J, CL, CP, NXT, OFF)                                    \
    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"                           \
    "ds_read_b64 " NXT ", v50 offset:" OFF "\n\t"                     \
    "s_waitcnt lgkmcnt(2)\n\t"                                        \
    "val" #J "_%=:\n\t"                                               \
    "v_mad_u64_u32 v[44:45], s[42:43], s40, " CL ", 0\n\t"            \
    "v_mad_u32_u24 v45, " CL ", s41, v45\n\t"                         \
    SEARCHTAIL(CP)                                                    \
    "s_cmp_eq_u32 s49, 0\n\t"                                         \
    "s_cbranch_scc1 rare_%=\n\t"                                      \
    "s_mov_b64 s[52:53], s[48:49]\n\t"                                \
    "s_sub_u32 s50, s50, s46\n\t"                                     \
    "s_subb_u32 s51, s51, s47\n\t"                                    \
    "v_writelane_b32 %[raw], s44, " #J "\n\t"

#endif
// speculative body of symbol J: (L, P) of its row's lane 7 in (SL, SP); reads the next row's into (NL, NP)
#define SBODY(J, NROWL, NROWP, NXT, OFF, SL, SP, NL, NP)              \
    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"                           \
    "ds_read_b64 " NXT ", v50 offset:" OFF "\n\t"                     \
    "s_mul_i32 s74, s40, " SL "\n\t"                                  \
    "s_mul_hi_u32 s75, s40, " SL "\n\t"                               \
    "s_mul_i32 s82, s41, " SL "\n\t"                                  \
    "s_add_u32 s75, s75, s82\n\t"                                     \
    "s_mul_i32 s76, s40, " SP "\n\t"                                  \
    "s_mul_hi_u32 s77, s40, " SP "\n\t"                               \
    "s_mul_i32 s82, s41, " SP "\n\t"                                  \
    "s_add_u32 s77, s77, s82\n\t"                                     \
    "s_waitcnt lgkmcnt(1)\n\t"                                        \
    "v_readlane_b32 " NL ", " NROWL ", 7\n\t"                         \
    "v_readlane_b32 " NP ", " NROWP ", 7\n\t"                         \
    "s_sub_u32 s78, s50, s74\n\t"                                     \
    "s_subb_u32 s79, s51, s75\n\t"                                    \
    "s_cbranch_scc1 val" #J "_%=\n\t"                                 \
    "s_sub_u32 s82, s78, s76\n\t"                                     \
    "s_subb_u32 s83, s79, s77\n\t"                                    \
    "s_cbranch_scc0 val" #J "_%=\n\t"                                 \
    "s_cmp_eq_u32 s77, 0\n\t"                                         \
    "s_cbranch_scc1 rare_%=\n\t"                                      \
    "s_mov_b64 s[50:51], s[78:79]\n\t"                                \
    "s_mov_b64 s[52:53], s[76:77]\n\t"

template <int SPEC>
__global__ __launch_bounds__(512) void dloop(uint64_t* out, int n_sym, int busy, int lead_hits) {
    __shared__ uint2 tab[20 * 64];
    __shared__ int junk[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 20 * 64; i += blockDim.x) {
        const int l = i & 63, row = i >> 6;
        const int hit = row < lead_hits ? 7 : 5;  // rows behind the leading hits are taken by lane 5: the mode test fails
        tab[i] = make_uint2(l < hit ? 0xffffffu : 0u, l == hit ? (1u << 24) - 127u : 0u);
    }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) junk[i] = i * 7 + 1;
    __syncthreads();
    if (wave == 0) {
        uint64_t rc_dist = 12345, rc_range = ~0ull;
        uint32_t status = 0;
        int raw = 0;
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        for (int rep = 0; rep < n_sym / 16; ++rep) {
            uint32_t taddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab)) + lane * 8;
            if (SPEC) {
                asm volatile(
                    "s_mov_b64 s[50:51], %[dst]\n\t"
                    "s_mov_b64 s[52:53], %[rng]\n\t"
                    "v_mov_b32 v50, %[ta]\n\t"
                    "v_mov_b32 %[raw], 7\n\t"
                    "ds_read_b64 v[40:41], v50\n\t"
                    "ds_read_b64 v[42:43], v50 offset:512\n\t"
                    "s_waitcnt lgkmcnt(1)\n\t"
                    "v_readlane_b32 s72, v40, 7\n\t"
                    "v_readlane_b32 s73, v41, 7\n\t"
                    SBODY(0, RL1, RP1, PAIR2, "1024", "s72", "s73", "s80", "s81")
                    SBODY(1, RL2, RP2, PAIR0, "1536", "s80", "s81", "s72", "s73")
                    SBODY(2, RL0, RP0, PAIR1, "2048", "s72", "s73", "s80", "s81")
                    SBODY(3, RL1, RP1, PAIR2, "2560", "s80", "s81", "s72", "s73")
                    SBODY(4, RL2, RP2, PAIR0, "3072", "s72", "s73", "s80", "s81")
                    SBODY(5, RL0, RP0, PAIR1, "3584", "s80", "s81", "s72", "s73")
                    SBODY(6, RL1, RP1, PAIR2, "4096", "s72", "s73", "s80", "s81")
                    SBODY(7, RL2, RP2, PAIR0, "4608", "s80", "s81", "s72", "s73")
                    SBODY(8, RL0, RP0, PAIR1, "5120", "s72", "s73", "s80", "s81")
                    SBODY(9, RL1, RP1, PAIR2, "5632", "s80", "s81", "s72", "s73")
                    SBODY(10, RL2, RP2, PAIR0, "6144", "s72", "s73", "s80", "s81")
                    SBODY(11, RL0, RP0, PAIR1, "6656", "s80", "s81", "s72", "s73")
                    SBODY(12, RL1, RP1, PAIR2, "7168", "s72", "s73", "s80", "s81")
                    SBODY(13, RL2, RP2, PAIR0, "7680", "s80", "s81", "s72", "s73")
                    SBODY(14, RL0, RP0, PAIR1, "8192", "s72", "s73", "s80", "s81")
                    SBODY(15, RL1, RP1, PAIR2, "8704", "s80", "s81", "s72", "s73")
                    "s_branch done_%=\n\t"
                    ".p2align 6\n\t"
                    VBODY(0, RL0, RP0, PAIR2, "1024")
                    VBODY(1, RL1, RP1, PAIR0, "1536")
                    VBODY(2, RL2, RP2, PAIR1, "2048")
                    VBODY(3, RL0, RP0, PAIR2, "2560")
                    VBODY(4, RL1, RP1, PAIR0, "3072")
                    VBODY(5, RL2, RP2, PAIR1, "3584")
                    VBODY(6, RL0, RP0, PAIR2, "4096")
                    VBODY(7, RL1, RP1, PAIR0, "4608")
                    VBODY(8, RL2, RP2, PAIR1, "5120")
                    VBODY(9, RL0, RP0, PAIR2, "5632")
                    VBODY(10, RL1, RP1, PAIR0, "6144")
                    VBODY(11, RL2, RP2, PAIR1, "6656")
                    VBODY(12, RL0, RP0, PAIR2, "7168")
                    VBODY(13, RL1, RP1, PAIR0, "7680")
                    VBODY(14, RL2, RP2, PAIR1, "8192")
                    VBODY(15, RL0, RP0, PAIR2, "8704")
                    "done_%=:\n\t"
                    "s_mov_b32 %[st], 0\n\t"
                    "s_branch out_%=\n\t"
                    "rare_%=:\n\t"
                    "s_mov_b32 %[st], 1\n\t"
                    "out_%=:\n\t"
                    "s_mov_b64 %[dst], s[50:51]\n\t"
                    "s_mov_b64 %[rng], s[52:53]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [raw] "+v"(raw), [st] "=s"(status)
                    : [ta] "v"(taddr), [lid] "v"(lane)
                    : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s72", "s73",
                      "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
                      "v49", "v50");
            } else {
                asm volatile(
                    "s_mov_b64 s[50:51], %[dst]\n\t"
                    "s_mov_b64 s[52:53], %[rng]\n\t"
                    "v_mov_b32 v50, %[ta]\n\t"
                    "v_mov_b32 %[raw], 7\n\t"
                    "ds_read_b64 v[40:41], v50\n\t"
                    "ds_read_b64 v[42:43], v50 offset:512\n\t"
                    ".p2align 6\n\t"
                    VBODY(0, RL0, RP0, PAIR2, "1024")
                    VBODY(1, RL1, RP1, PAIR0, "1536")
                    VBODY(2, RL2, RP2, PAIR1, "2048")
                    VBODY(3, RL0, RP0, PAIR2, "2560")
                    VBODY(4, RL1, RP1, PAIR0, "3072")
                    VBODY(5, RL2, RP2, PAIR1, "3584")
                    VBODY(6, RL0, RP0, PAIR2, "4096")
                    VBODY(7, RL1, RP1, PAIR0, "4608")
                    VBODY(8, RL2, RP2, PAIR1, "5120")
                    VBODY(9, RL0, RP0, PAIR2, "5632")
                    VBODY(10, RL1, RP1, PAIR0, "6144")
                    VBODY(11, RL2, RP2, PAIR1, "6656")
                    VBODY(12, RL0, RP0, PAIR2, "7168")
                    VBODY(13, RL1, RP1, PAIR0, "7680")
                    VBODY(14, RL2, RP2, PAIR1, "8192")
                    VBODY(15, RL0, RP0, PAIR2, "8704")
                    "s_mov_b32 %[st], 0\n\t"
                    "s_branch out_%=\n\t"
                    "rare_%=:\n\t"
                    "s_mov_b32 %[st], 1\n\t"
                    "out_%=:\n\t"
                    "s_mov_b64 %[dst], s[50:51]\n\t"
                    "s_mov_b64 %[rng], s[52:53]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [raw] "+v"(raw), [st] "=s"(status)
                    : [ta] "v"(taddr), [lid] "v"(lane)
                    : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "v40", "v41", "v42",
                      "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50");
            }
            if (status) break;
        }
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (lane < 16) out[16 + lane] = raw;
        if (lane == 0) { out[0] = t1 - t0; out[1] = rc_range; out[2] = status; out[4] = rc_dist; }
        junk[0] = 0x7fffffff;  // tell the busy waves to stop
    } else if (busy) {
        long long acc = wave;
        int idx = threadIdx.x;
        while (*reinterpret_cast<volatile int*>(&junk[0]) != 0x7fffffff) {
            for (int r = 0; r < 64; ++r) {
                const int4 w = *reinterpret_cast<const int4*>(&junk[((idx + r * 16) & 1020)]);
                acc += static_cast<long long>(w.x) * w.y + static_cast<long long>(w.z) * w.w;
                idx = (idx * 5 + 1) & 4095;
            }
        }
        if (acc == 42) out[8] = acc;
    }
}

int main() {
    uint64_t* d; (void)hipMalloc(&d, 512);
    const int n = 1 << 16;
    for (int lead : {16, 12, 9, 4, 1, 0}) {
        uint64_t h[2][32];
        double t[2][2];
        for (int spec = 0; spec < 2; ++spec)
            for (int busy = 0; busy < 2; ++busy) {
                for (int r = 0; r < 2; ++r) {
                    if (spec) hipLaunchKernelGGL(dloop<1>, dim3(1), dim3(512), 0, 0, d, n, busy, lead);
                    else hipLaunchKernelGGL(dloop<0>, dim3(1), dim3(512), 0, 0, d, n, busy, lead);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h[spec], d, 256, hipMemcpyDeviceToHost);
                t[spec][busy] = double(h[spec][0]) / n;
            }
        bool same = h[0][1] == h[1][1] && h[0][4] == h[1][4] && h[0][2] == 0 && h[1][2] == 0;
        for (int l = 0; l < 16; ++l) same = same && h[0][16 + l] == h[1][16 + l];
        printf("leading hits %2d / 16: search %6.1f (%6.1f busy)   speculative %6.1f (%6.1f busy) ticks / symbol   state + symbols %s\n", lead,
               t[0][0], t[0][1], t[1][0], t[1][1], same ? "identical" : "DIFFER");
    }
    return 0;
}
