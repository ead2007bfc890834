// The decoder's full-batch cycle in isolation: the generated 16-symbol block (cool_chic_amd/csrc/ccd_dec_block16.inc) + the
// hand-over of ccd_entropy_pipe.hip (symbols -> ring and latent grid, ready word reset, progress published, next batch's ready
// word / top symbols / rows requested, fast entry), over tables that are always ready.  What does a hand-over cost on top of the
// 16 symbols, alone and next to seven waves that keep the LDS busy?
//   V = 0: block + the least a loop needs (two row reads, compare, branch)      V = 1: + publication      V = 2: the production cycle
//     hipcc --offload-arch=gfx950 -O3 -I../../cool_chic_amd/csrc -o dcycle dcycle.hip && ./dcycle
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int V>
__global__ __launch_bounds__(512) void dcycle(uint64_t* out, int n_batches, int busy, int8_t* lat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: ring 32 KB at 0 | tables 128 rows x 64 x 8 B | top[128] | ready[16], consumed[2] | junk
    uint2* tab = reinterpret_cast<uint2*>(smem + 32768);
    int32_t* top = reinterpret_cast<int32_t*>(smem + 32768 + 65536);
    uint32_t* ready = reinterpret_cast<uint32_t*>(smem + 32768 + 65536 + 512);
    int* junk = reinterpret_cast<int*>(smem + 32768 + 65536 + 1024);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128 * 64; i += blockDim.x) {
        const int l = i & 63;
        tab[i] = make_uint2(l < 7 ? 0xffffffu : 0u, l == 7 ? (1u << 24) - 3u : 0u);  // lane 7 takes every symbol, the range barely shrinks
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) top[i] = 6;
    for (int i = threadIdx.x; i < 18; i += blockDim.x) ready[i] = i < 16 ? 3u : 0u;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) junk[i] = i * 7 + 1;
    __syncthreads();
    if (wave == 0) {
        uint64_t rc_dist = 12345, rc_range = ~0ull;
        uint32_t status = 0, i = 0, seq = 0, pix0 = 0, wpos = 2;
        const uint32_t n = static_cast<uint32_t>(n_batches) * 16u;
        int raw = 0, top_l = 0;
        uint32_t v_ring = static_cast<uint32_t>(lane << 6), v_goff = static_cast<uint32_t>(lane * 700);
        const uint32_t rdy = 32768 + 65536 + 512, tabl = 32768 + lane * 8, l4 = 32768 + 65536 + (lane & 15) * 4;
        const uint64_t lat_addr = reinterpret_cast<uint64_t>(lat);
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        asm volatile(
            "s_mov_b64 s[50:51], %[dst]\n\t"
            "s_mov_b64 s[52:53], %[rng]\n\t"
            "s_mov_b32 s69, 16\n\t"
            "s_mov_b32 s70, 7\n\t"
            "s_mov_b32 s71, 1024\n\t"
            "s_mov_b32 s72, %[n]\n\t"
            "s_mov_b32 s73, 3\n\t"
            "s_and_b32 s55, %[seq], %[smask]\n\t"
            "v_lshl_add_u32 v51, s55, 2, %[rdy]\n\t"
            "s_lshl_b32 s56, s55, 4\n\t"
            "v_lshl_add_u32 v53, s56, 2, %[l4]\n\t"
            "ds_read_b32 %[top], v53\n\t"
            "v_lshl_add_u32 v50, s56, 9, %[tabl]\n\t"
            "ds_read_b64 v[40:41], v50\n\t"
            "ds_read_b64 v[42:43], v50 offset:512\n\t"
            ".p2align 6\n\t"
            "80:\n\t"
#ifdef PAIRED
#include "ccd_dec_block16p.inc"
#else
#include "ccd_dec_block16.inc"
#endif
#if defined(VARIANT) && VARIANT >= 1
            "s_mov_b64 exec, 0xffff\n\t"
            "v_sub_u32 v52, %[top], %[raw]\n\t"
            "v_add_u32 v52, 1, v52\n\t"
            "ds_write_b8 %[ring], v52\n\t"
#ifndef NO_GLOBAL_STORE
            "global_store_byte %[goff], v52, %[lat]\n\t"
#endif
            "s_mov_b64 exec, -1\n\t"
            "ds_write_b32 v51, %[three]\n\t"       // (the production code writes 0; the producers that would set it again are not here)
            "s_add_u32 %[seq], %[seq], 1\n\t"
            "s_add_u32 s58, %[pix0], %[i]\n\t"
            "v_mov_b32 v56, %[seq]\n\t"
            "v_mov_b32 v57, s58\n\t"
            "ds_write_b64 %[rdy], v[56:57] offset:64\n\t"
#else
            "s_add_u32 %[seq], %[seq], 1\n\t"
#endif
#if defined(VARIANT) && VARIANT >= 2
            "s_and_b32 s55, %[seq], %[smask]\n\t"
            "v_lshl_add_u32 v51, s55, 2, %[rdy]\n\t"
            "ds_read_b32 v54, v51\n\t"
            "s_lshl_b32 s56, s55, 4\n\t"
            "v_lshl_add_u32 v53, s56, 2, %[l4]\n\t"
            "ds_read_b32 %[top], v53\n\t"
            "v_lshl_add_u32 v50, s56, 9, %[tabl]\n\t"
            "ds_read_b64 v[40:41], v50\n\t"
            "ds_read_b64 v[42:43], v50 offset:512\n\t"
            "s_add_u32 s54, %[i], s69\n\t"
            "s_cmp_le_u32 s54, s72\n\t"
            "s_cbranch_scc0 25f\n\t"
            "v_add_u32 %[ring], s71, %[ring]\n\t"
            "v_and_b32 %[ring], %[rmask], %[ring]\n\t"
            "v_add_u32 %[goff], %[gstride], %[goff]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_readfirstlane_b32 s59, v54\n\t"
            "s_cmp_eq_u32 s59, s73\n\t"
            "s_cbranch_scc1 80b\n\t"
            "25:\n\t"
#else
            "s_and_b32 s55, %[seq], %[smask]\n\t"
            "s_lshl_b32 s56, s55, 4\n\t"
            "v_lshl_add_u32 v50, s56, 9, %[tabl]\n\t"
            "ds_read_b64 v[40:41], v50\n\t"
            "ds_read_b64 v[42:43], v50 offset:512\n\t"
            "s_cmp_lt_u32 %[i], %[n]\n\t"
            "s_cbranch_scc1 80b\n\t"
#endif
            "s_mov_b32 %[st], 0\n\t"
            "s_branch 4f\n\t"
            "40:\n\t" "41:\n\t" "42:\n\t"
            "s_mov_b32 %[st], 1\n\t"
            "s_branch 4f\n\t"
            "15:\n\t"
            "s_mov_b32 %[st], 3\n\t"
            "s_branch 4f\n\t"
#ifdef PAIRED
#include "ccd_dec_tramp16p.inc"
#else
#include "ccd_dec_tramp16.inc"
#endif
            "4:\n\t"
            "s_mov_b64 %[dst], s[50:51]\n\t"
            "s_mov_b64 %[rng], s[52:53]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            : [dst] "+s"(rc_dist), [rng] "+s"(rc_range), [i] "+s"(i), [seq] "+s"(seq), [raw] "+v"(raw), [top] "+v"(top_l), [ring] "+v"(v_ring),
              [goff] "+v"(v_goff), [st] "=s"(status), [wpos] "+s"(wpos)
            : [n] "s"(n), [smask] "s"(7u), [rdy] "v"(rdy), [zero] "v"(0u), [three] "v"(3u), [rmask] "s"(511u * 64u + 63u), [gstride] "s"(64u),
              [tabl] "v"(tabl), [l4] "v"(l4), [lat] "s"(lat_addr), [pix0] "s"(pix0), [wbase] "s"(2u), [wbuf] "v"(static_cast<uint32_t>(lane * 2654435761u))
            : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58",
              "s59", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v56", "v57");
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[0] = t1 - t0; out[1] = rc_range; out[2] = status; out[3] = i; out[4] = rc_dist; }
        *reinterpret_cast<volatile int*>(&junk[0]) = 0x7fffffff;  // tell the busy waves to stop
    } else if (busy) {
        long long acc = wave;
        int idx = threadIdx.x;
        while (*reinterpret_cast<volatile int*>(&junk[0]) != 0x7fffffff) {
            for (int r = 0; r < 64; ++r) {
                const int4 w = *reinterpret_cast<const int4*>(&junk[((idx + r * 16) & 1020)]);
                acc += static_cast<long long>(w.x) * w.y + static_cast<long long>(w.z) * w.w;
                idx = (idx * 5 + 1) & 4095;
                if (idx == 0) idx = 4;
            }
        }
        if (acc == 42) out[8] = acc;
    }
}

int main() {
    uint64_t* d; int8_t* lat;
    hipMalloc(&d, 64 * 8); hipMalloc(&lat, 64 << 20);
    const int nb = 20000;
    const size_t lds = 32768 + 65536 + 1024 + 4096 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dcycle<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    for (int busy = 0; busy < 2; ++busy) {
        uint64_t h[8];
        for (int r = 0; r < 2; ++r) {
            hipLaunchKernelGGL(dcycle<VARIANT>, dim3(1), dim3(512), lds, 0, d, nb, busy, lat);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%svariant %d%s, %s: %.1f ticks per 16-symbol batch = %.1f / symbol   (status %llu, symbols %llu, range %016llx, dist %016llx)\n",
#ifdef PAIRED
               "paired test, ",
#else
               "",
#endif
               VARIANT,
#ifdef NO_GLOBAL_STORE
               " (no global store)",
#else
               "",
#endif
               busy ? "seven waves reading LDS" : "alone", double(h[0]) / nb, double(h[0]) / nb / 16, (unsigned long long)h[2], (unsigned long long)h[3], (unsigned long long)h[1], (unsigned long long)h[4]);
    }
    return 0;
}
