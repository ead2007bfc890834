// The decoder's symbol loop (ccd_entropy_pipe.hip, decoder_grid) with and without the VALU -> SALU -> VALU hop of the lane
// search: variant 0 = what the kernel runs (v_cmp -> vcc -> s_ff1 -> four SGPR-indexed v_readlane), variant 1 = v_cmpx
// writes EXEC (and VCC) and four v_readfirstlane read the hit lane directly; EXEC is restored with one s_mov, the lane
// index (needed only for the symbol bookkeeping) comes from s_ff1 on VCC off the chain.  One wave alone on its CU and
// next to 7 busy waves, like dloop.hip.      hipcc --offload-arch=gfx950 -O3 -o dloop_cmpx dloop_cmpx.hip && ./dloop_cmpx
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define HEAD(CURL)                                                     \
    "s_lshr_b64 s[40:41], s[52:53], 24\n\t"
#define SEARCH0(CURL, CURP)                                            \
    "v_mad_u64_u32 v[44:45], s[42:43], s40, " CURL ", 0\n\t"           \
    "v_mad_u32_u24 v45, " CURL ", s41, v45\n\t"                        \
    "v_cmp_ge_u64 vcc, s[50:51], v[44:45]\n\t"                         \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CURP ", 0\n\t"           \
    "v_mad_u32_u24 v49, " CURP ", s41, v49\n\t"                        \
    "s_and_b32 m0, %[i], 15\n\t"                                       \
    "s_ff1_i32_b64 s44, vcc\n\t"                                       \
    "v_readlane_b32 s48, v48, s44\n\t"                                 \
    "v_readlane_b32 s49, v49, s44\n\t"                                 \
    "v_readlane_b32 s46, v44, s44\n\t"                                 \
    "v_readlane_b32 s47, v45, s44\n\t"
#define SEARCH1(CURL, CURP)                                            \
    "v_mad_u64_u32 v[44:45], s[42:43], s40, " CURL ", 0\n\t"           \
    "v_mad_u32_u24 v45, " CURL ", s41, v45\n\t"                        \
    "v_cmpx_ge_u64 vcc, s[50:51], v[44:45]\n\t"                        \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CURP ", 0\n\t"           \
    "v_mad_u32_u24 v49, " CURP ", s41, v49\n\t"                        \
    "s_and_b32 m0, %[i], 15\n\t"                                       \
    "s_ff1_i32_b64 s44, vcc\n\t"                                       \
    "v_readfirstlane_b32 s48, v48\n\t"                                 \
    "v_readfirstlane_b32 s49, v49\n\t"                                 \
    "v_readfirstlane_b32 s46, v44\n\t"                                 \
    "v_readfirstlane_b32 s47, v45\n\t"                                 \
    "s_mov_b64 exec, -1\n\t"
// variant 2: like 1, but scale * P is computed for all lanes BEFORE the compare (no dependence of its issue slot on EXEC)
#define SEARCH2(CURL, CURP)                                            \
    "v_mad_u64_u32 v[44:45], s[42:43], s40, " CURL ", 0\n\t"           \
    "v_mad_u64_u32 v[48:49], s[42:43], s40, " CURP ", 0\n\t"           \
    "v_mad_u32_u24 v45, " CURL ", s41, v45\n\t"                        \
    "v_mad_u32_u24 v49, " CURP ", s41, v49\n\t"                        \
    "v_cmpx_ge_u64 vcc, s[50:51], v[44:45]\n\t"                        \
    "s_nop 3\n\t"                                                      \
    "v_readfirstlane_b32 s48, v48\n\t"                                 \
    "v_readfirstlane_b32 s49, v49\n\t"                                 \
    "v_readfirstlane_b32 s46, v44\n\t"                                 \
    "v_readfirstlane_b32 s47, v45\n\t"                                 \
    "s_mov_b64 exec, -1\n\t"                                           \
    "s_and_b32 m0, %[i], 15\n\t"                                       \
    "s_ff1_i32_b64 s44, vcc\n\t"
#define TAILC                                                          \
    "s_cmp_eq_u32 s49, 0\n\t"                                          \
    "s_cbranch_scc1 3f\n\t"                                            \
    "s_mov_b64 s[52:53], s[48:49]\n\t"                                 \
    "s_sub_u32 s50, s50, s46\n\t"                                      \
    "s_subb_u32 s51, s51, s47\n\t"                                     \
    "v_writelane_b32 %[raw], s44, m0\n\t"                              \
    "s_add_u32 %[i], %[i], 1\n\t"

#define LOOP(SEARCH)                                                                                      \
    "s_mov_b64 s[50:51], %[dst]\n\t"                                                                      \
    "s_mov_b64 s[52:53], %[rng]\n\t"                                                                      \
    "ds_read_b64 v[40:41], %[ta]\n\t"                                                                     \
    "ds_read_b64 v[42:43], %[ta] offset:512\n\t"                                                          \
    ".p2align 6\n\t"                                                                                      \
    "1:\n\t"                                                                                              \
    HEAD("v40") "ds_read_b64 v[46:47], %[ta] offset:1024\n\ts_waitcnt lgkmcnt(2)\n\t" SEARCH("v40", "v41") TAILC "s_cmp_lt_u32 %[i], %[cnt]\n\ts_cbranch_scc0 2f\n\t" \
    HEAD("v42") "ds_read_b64 v[40:41], %[ta] offset:1536\n\ts_waitcnt lgkmcnt(2)\n\t" SEARCH("v42", "v43") TAILC "s_cmp_lt_u32 %[i], %[cnt]\n\ts_cbranch_scc0 2f\n\t" \
    HEAD("v46") "ds_read_b64 v[42:43], %[ta] offset:2048\n\ts_waitcnt lgkmcnt(2)\n\t" SEARCH("v46", "v47") TAILC "v_add_u32 %[ta], 0x600, %[ta]\n\ts_cmp_lt_u32 %[i], %[cnt]\n\ts_cbranch_scc1 1b\n\t" \
    "2:\n\t"                                                                                              \
    "s_mov_b32 %[st], 0\n\t"                                                                              \
    "s_branch 4f\n\t"                                                                                     \
    "3:\n\t"                                                                                              \
    "s_mov_b64 exec, -1\n\t"                                                                              \
    "s_mov_b32 %[st], 1\n\t"                                                                              \
    "4:\n\t"                                                                                              \
    "s_mov_b32 %[kr], s44\n\t"                                                                            \
    "s_mov_b64 %[dst], s[50:51]\n\t"                                                                      \
    "s_mov_b64 %[rng], s[52:53]\n\t"                                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"

template <int VARIANT>
__global__ __launch_bounds__(512) void dloop(uint64_t* out, int n_sym, int busy, int hit_lane) {
    __shared__ uint2 tab[20 * 64];
    __shared__ int junk[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 20 * 64; i += blockDim.x) {
        const int l = i & 63;  // lane `hit_lane` always hits, range barely shrinks
        tab[i] = make_uint2(l < hit_lane ? 0xffffffu : 0u, l == hit_lane ? (1u << 24) - 127u : 0u);
    }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) junk[i] = i * 7 + 1;
    __syncthreads();
    if (wave == 0) {
        uint64_t rc_dist = 12345, rc_range = ~0ull;
        uint32_t i = 0, status = 0, k_rare = 0;
        int raw = 0;
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        for (int rep = 0; rep < n_sym / 16; ++rep) {
            uint32_t taddr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tab)) + lane * 8;
            i = 0;
            const uint32_t cnt = 16;
#define ASM(SEARCH)                                                                                                                         \
            asm volatile(LOOP(SEARCH)                                                                                                        \
                : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [i] "+s"(i), [raw] "+v"(raw), [ta] "+v"(taddr), [st] "=s"(status), [kr] "=s"(k_rare) \
                : [cnt] "s"(cnt)                                                                                                            \
                : "memory", "vcc", "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",    \
                  "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49")
            if (VARIANT == 0) ASM(SEARCH0);
            else if (VARIANT == 1) ASM(SEARCH1);
            else ASM(SEARCH2);
            if (status) break;
        }
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[0] = t1 - t0; out[1] = rc_range; out[2] = status; out[3] = raw; out[4] = rc_dist; out[5] = k_rare; }
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

template <int V>
static void run(uint64_t* d, const char* name) {
    for (int busy = 0; busy < 2; ++busy) {
        uint64_t h[6];
        const int n = 1 << 16;
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(dloop<V>, dim3(1), dim3(512), 0, 0, d, n, busy, 3); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
        printf("%-28s %s: %6.1f ticks / symbol (status %llu, range %016llx, dist %016llx, raw lane0 %llu, k %llu)\n", name, busy ? "7 busy waves" : "alone       ",
               double(h[0]) / n, (unsigned long long)h[2], (unsigned long long)h[1], (unsigned long long)h[4], (unsigned long long)h[3], (unsigned long long)h[5]);
    }
}
int main() {
    uint64_t* d; (void)hipMalloc(&d, 256);
    run<0>(d, "v_cmp + s_ff1 + v_readlane");
    run<1>(d, "v_cmpx + v_readfirstlane");
    run<2>(d, "v_cmpx, products first");
    return 0;
}
