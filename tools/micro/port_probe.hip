// Micro-benchmark (diagnostic, not part of the product): what one CU's vector-memory path delivers into LDS.
// One workgroup per CU (150 KiB of LDS forces that), NL loader waves streaming 1 KiB pieces into an LDS ring with a counted
// vmcnt throttle, optionally beside NC waves that issue bare MFMAs + ds_read_b128.  Sources: a per-workgroup private HBM
// region (read once), a 256 KiB buffer every workgroup re-reads (L2), or both alternating.  Methods: LDS-DMA
// (global_load_lds_dwordx4), register staging (global_load_dwordx4 + ds_write_b128), and a "touch" (one byte of each of
// 64 different 128-B lines per wave-instruction: does a cheap touch pull lines into L2 ahead of the real read?).
// Prints cycles per 1 KiB piece per CU (median over workgroups, s_memtime) and the chip rate from the wall clock.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
    const char* hbm; size_t per_wg;     // private region per workgroup
    const char* l2; int l2_bytes;       // shared buffer
    unsigned long long* cyc;            // [nwg]
    float* sink;
    int pieces;                         // 1 KiB pieces per loader wave
    int nl, nc;                         // loader / compute waves
    int src;                            // 0 hbm, 1 l2, 2 alternate, 3 hbm default policy (no nt)
    int method;                         // 0 LDS-DMA, 1 register staging, 2 touch then DMA
    int depth;                          // pieces in flight per loader wave
};

template <int DEPTH> __device__ __forceinline__ void wait_depth() {
    if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (DEPTH == 32) asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(55)" ::: "memory");
}

template <int DEPTH, int SRC, int METHOD>
__global__ void __launch_bounds__(1024) k(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long t0 = 0;
    if (tid == 0) *reinterpret_cast<unsigned long long*>(smem + 150 * 1024 - 8) = 0ull;
    __syncthreads();
    if (wave == 0 && lane == 0) t0 = __builtin_amdgcn_s_memtime();
    if (wave < p.nl) {
        // ---- loader: the loaders share a 128-slot ring, slot = (piece * nl + wave) & 127; all address arithmetic is incremental
        const char* gh = p.hbm + (size_t)blockIdx.x * p.per_wg + (size_t)wave * p.pieces * 1024 + lane * 16;
        const char* gl = p.l2 + lane * 16;
        unsigned l2off = wave * 1024u, slot = wave;
        const unsigned l2mask = p.l2_bytes - 1, step = p.nl * 1024u, nl = p.nl;
        auto piece = [&](auto L2c) {
            constexpr bool from_l2 = decltype(L2c)::value;
            char* dst = smem + ((slot & 127u) << 10);
            slot += nl;
            if constexpr (METHOD == 0 || METHOD == 2) {
                if constexpr (from_l2) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gl + (l2off & l2mask)), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    l2off += step;
                } else {
                    if constexpr (SRC == 3) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gh, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gh, (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
                    gh += 1024;
                }
                wait_depth<DEPTH>();
            }
        };
        if constexpr (METHOD == 1) {
            // register staging, 8 pieces per batch: 8 loads in flight, then 8 ds_write_b128 (the compiler's own counted waits)
            for (int i = 0; i < p.pieces; i += 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool l2j = SRC == 1 || (SRC == 2 && (j & 1));
                    if (l2j) { v[j] = *reinterpret_cast<const u32x4*>(gl + (l2off & l2mask)); l2off += step; }
                    else { v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gh)); gh += 1024; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { *reinterpret_cast<u32x4*>(smem + ((slot & 127u) << 10) + lane * 16) = v[j]; slot += nl; }
            }
        } else {
            for (int i = 0; i < p.pieces; i += 2) {
                if constexpr (METHOD == 2) {
                    // touch 64 lines (8 KiB = 8 pieces) 64 pieces ahead, once every 8 pieces; destination v100: above what this kernel
                    // allocates (checked in the disassembly), never read
                    if ((i & 7) == 0 && i + 64 + 8 <= p.pieces) {
                        const char* tp = gh + 64 * 1024 + lane * 112;      // gh already carries lane * 16: lane * 128 in all
                        asm volatile("global_load_ubyte v100, %0, off" : : "v"(tp) : "memory", "v100");
                    }
                }
                piece(std::integral_constant<bool, SRC == 1>{});
                piece(std::integral_constant<bool, SRC == 1 || SRC == 2>{});
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wave < p.nl + p.nc) {
        // ---- compute: bare MFMAs with one ds_read_b128 per two MFMAs, sized to outlast the loaders roughly
        f32x4 a[6];
        for (int j = 0; j < 6; ++j) a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* rb = smem + 128 * 1024 + (wave & 7) * 2048;
        const int n = p.pieces * p.nl / 4;
        for (int i = 0; i < n; ++i) {
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(rb + lane * 16);
            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(rb + 1024 + lane * 16);
            const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(rb + ((i & 1) << 10) + lane * 16);
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? x0 : j < 4 ? x1 : x2, x1, a[j], 0, 0, 0);
        }
        float s = 0.f;
        for (int j = 0; j < 6; ++j) s += a[j][0];
        if (s == 12345.f) p.sink[1] = s;
    }
    if (wave < p.nl && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        atomicMax(reinterpret_cast<unsigned long long*>(smem + 150 * 1024 - 8), t1);
    }
    __syncthreads();
    if (wave == 0 && lane == 0) p.cyc[blockIdx.x] = *reinterpret_cast<unsigned long long*>(smem + 150 * 1024 - 8) - t0;
}

typedef void (*kern_t)(P);
template <int SRC, int METHOD> kern_t pick_depth(int depth) {
    return depth == 4 ? k<4, SRC, METHOD> : depth == 8 ? k<8, SRC, METHOD> : depth == 16 ? k<16, SRC, METHOD> : depth == 32 ? k<32, SRC, METHOD> : k<56, SRC, METHOD>;
}
template <int METHOD> kern_t pick_src(int src, int depth) {
    return src == 0 ? pick_depth<0, METHOD>(depth) : src == 1 ? pick_depth<1, METHOD>(depth) : src == 2 ? pick_depth<2, METHOD>(depth) : pick_depth<3, METHOD>(depth);
}

int main(int argc, char** argv) {
    int ncu = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    if (prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
    const size_t per_wg = 8u << 20;
    char *hbm, *l2;
    unsigned long long* cyc;
    float* sink;
    CHECK(hipMalloc(&hbm, per_wg * ncu));
    CHECK(hipMemset(hbm, 1, per_wg * ncu));
    CHECK(hipMalloc(&l2, 256 << 10));
    CHECK(hipMemset(l2, 1, 256 << 10));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * ncu));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t ea, eb;
    CHECK(hipEventCreate(&ea));
    CHECK(hipEventCreate(&eb));
    const int lds = 150 * 1024;
    auto run = [&](int nl, int nc, int src, int method, int depth, int nwg) {
        P p{hbm, per_wg, l2, 256 << 10, cyc, sink, 0, nl, nc, src, method, depth};
        p.pieces = (int)std::min<size_t>(per_wg / 1024 / nl, 2048);
        kern_t kern = method == 0 ? pick_src<0>(src, depth) : method == 1 ? pick_src<1>(src, depth) : pick_src<2>(src, depth);
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        float best = 1e30f;
        std::vector<unsigned long long> h(nwg);
        for (int it = 0; it < 3; ++it) {
            CHECK(hipEventRecord(ea));
            hipLaunchKernelGGL(kern, dim3(nwg), dim3((nl + nc) * 64), lds, 0, p);
            CHECK(hipEventRecord(eb));
            CHECK(hipEventSynchronize(eb));
            float ms;
            CHECK(hipEventElapsedTime(&ms, ea, eb));
            best = std::min(best, ms);
        }
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nwg, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double npieces = (double)p.pieces * nl;
        const double med = (double)h[nwg / 2];
        printf("nl=%2d nc=%2d src=%d method=%d depth=%2d nwg=%3d | %7.1f cyc/piece/CU  %6.2f B/clk/CU | wall %8.3f ms  chip %6.2f TB/s\n", nl, nc, src, method, depth, nwg,
               med / npieces, 1024.0 * npieces / med, best, npieces * 1024.0 * nwg / best / 1e9);
        fflush(stdout);
    };
    const char* names[] = {"HBM nt", "L2 shared 256 KiB", "alternating HBM nt / L2", "HBM default policy"};
    for (int src = 0; src < 4; ++src) {
        printf("## LDS-DMA, source: %s, loaders alone\n", names[src]);
        for (int nl : {1, 2, 4, 8, 12})
            for (int depth : {4, 16, 56}) run(nl, 0, src, 0, depth, ncu);
    }
    printf("## LDS-DMA beside 8 MFMA waves\n");
    for (int src = 0; src < 3; ++src)
        for (int nl : {2, 4})
            for (int depth : {8, 16, 32}) run(nl, 8, src, 0, depth, ncu);
    printf("## register staging (global_load_dwordx4 + ds_write_b128), loaders alone\n");
    for (int src = 0; src < 3; ++src)
        for (int nl : {4, 8}) run(nl, 0, src, 1, 8, ncu);
    printf("## touch 64 pieces ahead (one byte per 128-B line), then LDS-DMA nt from HBM\n");
    for (int nl : {1, 2, 4})
        for (int depth : {8, 16, 32}) run(nl, 0, 0, 2, depth, ncu);
    printf("## 8 workgroups only (is it a per-CU or a chip limit?)\n");
    for (int src = 0; src < 3; ++src)
        for (int depth : {8, 32}) run(4, 0, src, 0, depth, 8);
    return 0;
}
