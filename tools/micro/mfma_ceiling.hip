// Micro-benchmark (diagnostic, not part of the product): what the matrix cores sustain on this chip on RANDOM bf16 data, at the clock the
// chip holds under that load -- the practical ceiling the fused conv kernels are measured against beside the nominal 2.5 PFLOP/s.
//   mode 0: bare v_mfma_f32_16x16x32_bf16, operands in registers, 18 independent accumulators (the dense-block kernels' shape)
//   mode 1: the same MFMA stream with its operands re-read from LDS by ds_read_b128 in the fused kernels' ratio (30 reads per 54 MFMAs)
//   mode 2: mode 1 + four loader waves streaming 29 KiB per 54-MFMA granule into LDS by LDS-DMA (11 KiB from HBM, 18 KiB from L2)
//   mode 8 (round 4): mode 2 with the kx-shifted pixel fragments made in registers instead of re-read from LDS: 4 instead of 12 pixel-fragment reads per granule, and
//           72 v_mov_b32_dpp (row_ror:1 of the neighbour column group's fragment, then row_shr:1 of the own one over it) -- what a kernel that walks a chunk's three
//           kx taps on ONE staging of the rows would issue
//   mode 6 / 7 (round 4): the same flops on v_mfma_f32_32x32x16_bf16 -- a 32-cout x 32-pixel tile per instruction, HALF the operand bytes (registers and LDS) per flop:
//           four compute waves (one per SIMD), nine 32 x 32 accumulator tiles each (conv4 + conv5 of two image rows); 6 = bare, 7 = 30 LDS reads per 54 MFMAs + the
//           loaders' 29 KiB of LDS-DMA + one barrier per granule (what mode 2 is to the kernels as they are, this is to a kernel rebuilt on 32 x 32 tiles)
// One workgroup per CU (150 KiB of LDS), 8 compute waves (two per SIMD).  Each configuration runs back to back for ~2 s before it is timed;
// the in-kernel clock is s_memtime / s_memrealtime (100 MHz) over the timed launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
    const char* rnd;          // 64 KiB of random bf16 (finite, |x| < 2)
    const char* hbm; size_t per_wg;
    unsigned long long* stamps;   // [nwg][4]: memtime0, realtime0, memtime1, realtime1
    float* sink;
    int iters;                // granules (54 MFMAs per compute wave each)
};

template <int MODE>
__global__ void __launch_bounds__((MODE == 2 || MODE == 4 || MODE == 5 || MODE == 8) ? 768 : (MODE == 6 ? 256 : 512)) k(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS: [0, 64 KiB) random operands, [64, 128 KiB) loader ring
    for (int u = tid; u < 4096; u += blockDim.x) reinterpret_cast<u32x4*>(smem)[u] = reinterpret_cast<const u32x4*>(p.rnd)[u];
    unsigned* flags = reinterpret_cast<unsigned*>(smem + 140 * 1024);      // [0] granules published x 4 loaders, [1] granules finished x 8 compute waves
    if (tid < 2) flags[tid] = tid == 0 ? 4u : 0u;                           // granule 0's operands are the random block already in LDS
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if ((MODE == 6 || MODE == 7) && wave < 4) {
        f32x16 acc[9];
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const char* base = smem + wave * 4096 + lane * 16;
        bf16x8 w[4], x[2][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(base + j * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) { x[0][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + j * 1024); x[1][j] = *reinterpret_cast<const bf16x8*>(base + 28672 + j * 1024); }
        for (int it = 0; it < p.iters; ++it) {
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (MODE == 7) {
                    // one weight fragment per stage (3 stages ahead), three pixel fragments every 4-5 stages: 18 + 12 reads per 54 MFMAs (two rows' worth of pixels)
                    w[(s + 3) & 3] = *reinterpret_cast<const bf16x8*>(base + (((s + it) & 31) << 10));
                    if (s == 0 || s == 4 || s == 10 || s == 16) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) x[(s >> 2) & 1][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + (((s + j + it) & 7) << 10));
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[(s % 3) * 3 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s & 3], x[((s + 2) >> 2) & 1][nt], acc[(s % 3) * 3 + nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) sum += acc[j][0] + acc[j][15];
        if (sum == 12345.678f) p.sink[0] = sum;
    } else if (MODE == 3 && wave < 4) {
        // one compute wave per SIMD, two image rows per wave: every weight fragment meets six pixel fragments
        f32x4 acc[36];
#pragma unroll
        for (int j = 0; j < 36; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* base = smem + wave * 4096 + lane * 16;
        bf16x8 w[4], x[3][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(base + j * 1024);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) x[r][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + (r * 3 + j) * 1024);
        for (int it = 0; it < p.iters; ++it) {
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                w[(s + 3) & 3] = *reinterpret_cast<const bf16x8*>(base + (((s + it) & 31) << 10));
                if (s == 0 || s == 4 || s == 8 || s == 12 || s == 16) {         // five pixel rows per granule
#pragma unroll
                    for (int j = 0; j < 3; ++j) x[(s >> 2) % 3][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + (((s + j + it) & 7) << 10));
                }
#pragma unroll
                for (int cg = 0; cg < 3; ++cg) {
                    acc[2 * s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s & 3], x[((s + 2) >> 2) % 3][cg], acc[2 * s], 0, 0, 0);
                    acc[2 * s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s & 3], x[((s + 6) >> 2) % 3][cg], acc[2 * s + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 36; ++j) sum += acc[j][0] + acc[j][3];
        if (sum == 12345.678f) p.sink[0] = sum;
    } else if (MODE != 3 && MODE != 6 && MODE != 7 && wave < 8) {
        f32x4 acc[18];
#pragma unroll
        for (int j = 0; j < 18; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* base = smem + wave * 4096 + lane * 16;
        bf16x8 w[4], x[2][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(base + j * 1024);
#pragma unroll
        for (int j = 0; j < 3; ++j) { x[0][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + j * 1024); x[1][j] = *reinterpret_cast<const bf16x8*>(base + 28672 + j * 1024); }
        // mode 5: two barriers per granule, waves 4-7 (the SIMD partners of waves 0-3) run half a granule behind: one extra barrier up front,
        // one more at the end (MI355X_MICROARCH.md "try a stagger"): a wave's barrier wait and post-barrier ramp meet its partner's MFMAs
        if (MODE == 5 && wave >= 4) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < p.iters; ++it) {
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (MODE == 5 && s == 9) __builtin_amdgcn_s_barrier();      // mid-granule: the partner group's boundary; this wave's reads stay in flight
                if (MODE == 8) {
                    w[(s + 3) & 3] = *reinterpret_cast<const bf16x8*>(base + (((s + it) & 31) << 10));
                    if (s == 0) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) x[0][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + (((s + j + it) & 7) << 10));
                    }
                    if (s == 10) x[1][0] = *reinterpret_cast<const bf16x8*>(base + 24576 + (((s + it) & 7) << 10));
                    {
                        // two dwords of one fragment shifted by a pixel: lane 0 of a 16-lane row from the neighbour column group's lane 15, the rest from the own fragment
                        typedef int i32x4 __attribute__((ext_vector_type(4)));
                        i32x4 own = __builtin_bit_cast(i32x4, x[(s >> 2) & 1][s % 3]), nb = __builtin_bit_cast(i32x4, x[(s >> 2) & 1][(s + 2) % 3]);
                        i32x4 o = __builtin_bit_cast(i32x4, x[((s >> 2) + 1) & 1][s % 3]);
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int d = 2 * (s & 1) + a;
                            const int t = __builtin_amdgcn_mov_dpp(nb[d], 0x121, 0xf, 0xf, false);              // row_ror:1
                            o[d] = __builtin_amdgcn_update_dpp(t, own[d], 0x111, 0xf, 0xf, false);                // row_shr:1, lane 0 keeps t
                        }
                        x[((s >> 2) + 1) & 1][s % 3] = __builtin_bit_cast(bf16x8, o);
                    }
                } else if (MODE >= 1) {
                    // one weight fragment per stage (3 stages ahead), three pixel fragments every 4-5 stages: 18 + 12 reads per 54 MFMAs
                    w[(s + 3) & 3] = *reinterpret_cast<const bf16x8*>(base + (((s + it) & 31) << 10));
                    if (s == 0 || s == 4 || s == 10 || s == 16) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) x[(s >> 2) & 1][j] = *reinterpret_cast<const bf16x8*>(base + 24576 + (((s + j + it) & 7) << 10));
                    }
                }
#pragma unroll
                for (int cg = 0; cg < 3; ++cg)
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s & 3], x[((s + 2) >> 2) & 1][cg], acc[s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2 || MODE == 5 || MODE == 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            if (MODE == 4) {
                // done: this wave has finished reading granule `it`; then wait until the loaders have published granule it + 1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spins = 0;
                while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4 * (it + 2) && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            }
        }
        if (MODE == 5 && wave < 4) __builtin_amdgcn_s_barrier();
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 18; ++j) sum += acc[j][0] + acc[j][3];
        if (sum == 12345.678f) p.sink[0] = sum;
    } else if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8) {
        // loaders: per granule 29 pieces over 4 waves (7-8 each): ~11 from a private HBM stream (nt), ~18 from the shared random buffer (L2)
        const int lw = wave - ((MODE == 3 || MODE == 7) ? 4 : 8);
        (void)flags;
        const char* gh = p.hbm + (size_t)blockIdx.x * p.per_wg + (size_t)lw * (p.per_wg / 4) + lane * 16;
        const char* gl = p.rnd + lane * 16;
        unsigned slot = lw;
        const size_t hmask = p.per_wg / 4 - 1;
        size_t ho = 0;
        for (int it = 0; it < p.iters; ++it) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                char* dst = smem + 65536 + ((slot & 63u) << 10);
                slot += 4;
                if (j < 3) { __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gh + (ho & hmask)), (__attribute__((address_space(3))) void*)dst, 16, 0, 2); ho += 1024; }
                else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gl + (((it * 7 + j) & 63) << 10)), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if (MODE == 4) {
                // publish (each loader counts once per granule: ready = 4 (it + 1) when all four have), then wait until every compute wave is
                // done with granule it - 1 before the ring slot it used is overwritten two iterations on
                if (lane == 0) __hip_atomic_fetch_add(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spins = 0;
                while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 8 * (it - 1) && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            } else {
                __builtin_amdgcn_s_barrier();
                if (MODE == 5) __builtin_amdgcn_s_barrier();
            }
        }
        if (MODE == 5) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        p.stamps[blockIdx.x * 4 + 0] = t0; p.stamps[blockIdx.x * 4 + 1] = r0;
        p.stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    if (prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    std::vector<unsigned short> h(32768);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const unsigned e = 120 + ((s >> 8) & 7); v = (unsigned short)(((s >> 31) << 15) | (e << 7) | ((s >> 12) & 0x7f)); }   // |x| in [2^-7, 2)
    char *rnd, *hbm;
    unsigned long long* stamps;
    float* sink;
    const size_t per_wg = 4u << 20;
    CHECK(hipMalloc(&rnd, 65536)); CHECK(hipMemcpy(rnd, h.data(), 65536, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&hbm, per_wg * ncu)); CHECK(hipMemset(hbm, 0x3c, per_wg * ncu));
    CHECK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * ncu));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t ea, eb;
    CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
    const int lds = 150 * 1024;
    const char* names[] = {"bare MFMA, operands in registers", "MFMA + LDS operand reads (30 ds_read_b128 per 54 MFMAs)", "MFMA + LDS reads + 29 KiB of LDS-DMA per granule + one barrier per granule",
                           "4 compute waves (one per SIMD) x two rows: 33 reads per 108 MFMAs, + LDS-DMA + barrier per granule",
                           "mode 2 with the per-granule barrier replaced by a ready / done handshake through LDS counters",
                           "mode 2 with two barriers per granule and waves 4-7 half a granule behind their SIMD partners (stagger)",
                           "bare v_mfma_f32_32x32x16_bf16, 4 waves (one per SIMD) x nine 32x32 tiles, operands in registers",
                           "32x32x16 MFMAs, 4 compute waves: 30 ds_read_b128 per 54 MFMAs (half of mode 2's per flop) + 29 KiB of LDS-DMA + one barrier per granule",
                           "mode 2 with 22 LDS reads per granule and 72 v_mov_b32_dpp making the kx-shifted pixel fragments in registers"};
    const int only = getenv("MFMA_MODE") ? atoi(getenv("MFMA_MODE")) : -1;
    for (int mode = 0; mode < 9; ++mode) {
        if (only >= 0 && mode != only) continue;
        auto kern = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : mode == 5 ? k<5> : mode == 6 ? k<6> : mode == 7 ? k<7> : k<8>;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        P p{rnd, hbm, per_wg, stamps, sink, 20000};
        const int threads = (mode == 2 || mode == 4 || mode == 5 || mode == 8) ? 768 : (mode == 6 ? 256 : 512);      // modes 3, 7: 4 compute + 4 loader waves
        float ms = 0.f, total = 0.f;
        int n = 0;
        while (total < 2500.f && n < 400) {             // ~2.5 s of back-to-back launches, the last one is the measurement
            CHECK(hipEventRecord(ea));
            hipLaunchKernelGGL(kern, dim3(ncu), dim3(threads), lds, 0, p);
            CHECK(hipEventRecord(eb));
            CHECK(hipEventSynchronize(eb));
            CHECK(hipEventElapsedTime(&ms, ea, eb));
            total += ms; ++n;
        }
        std::vector<unsigned long long> st(4 * ncu);
        CHECK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * 4 * ncu, hipMemcpyDeviceToHost));
        std::vector<double> clk, cyc;
        for (int i = 0; i < ncu; ++i) {
            const double dt = (double)(st[i * 4 + 2] - st[i * 4 + 0]), dr = (double)(st[i * 4 + 3] - st[i * 4 + 1]);
            if (dr > 0) { clk.push_back(dt / dr * 100.0); cyc.push_back(dt); }
        }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        const double flop = (double)ncu * 8 * 54.0 * p.iters * 16 * 16 * 32 * 2;      // mode 3: 4 waves x 108 MFMAs = the same
        const double mfma_per_simd = 2.0 * 54.0 * p.iters;      // in units of 16-cycle MFMAs (modes 6, 7: one wave per SIMD, 54 32-cycle MFMAs)
        printf("mode %d (%s): %.3f ms, %.1f TFLOP/s, in-kernel clock %.0f MHz (median), %.2f cycles per MFMA per SIMD, pipe utilisation %.3f, %d launches\n", mode, names[mode], ms,
               flop / ms / 1e9, clk[clk.size() / 2], cyc[cyc.size() / 2] / mfma_per_simd, 16.0 * mfma_per_simd / cyc[cyc.size() / 2], n);
        fflush(stdout);
    }
    return 0;
}
