// conv_stream.hip -- bf16 3x3 SAME convolution from 64 input channels to 64 couts per workgroup (NHWC in, any conv_rows epilogue out) as a PERSISTENT
// kernel with the cout tile's weights resident in LDS.
//
// Why.  On the tile kernel (conv_rows.hip) a 64 -> 64 conv stages 72 KiB of weights for every 12 x 16 output tile -- 1.1 KB per pixel through
// L2 -> LDS against 0.5 KB of activations -- and the layers that have only these two chunks (the generator's up-sampling convs,
// EDSR's whole body, VGG16's block1_conv2) run at 0.65-0.9 PFLOP/s.  Here a workgroup keeps ONE cout tile's 72 KiB (both chunks, all nine taps) for
// its whole life and walks a contiguous run of 24 x 16 output tiles:
//   * roles as in dense_fused.hip: eight compute waves (three output rows of the tile each, the row-sliding MFMA loop of conv_rows.hip) + four
//     loader waves that own the LDS-DMA stream with counted vmcnt; one barrier per 32-channel chunk;
//   * unit k = (tile k / 2, chunk k % 2) lands in buffer k % 3 (26 x 18 halo pixels x 64 B, the row kernels' XOR-swizzled image: the swizzle is on
//     the DMA's source side); the loaders run two units ahead, so the next tile's first chunk flies under this tile's second;
//   * out-of-image halo pixels come from a zero page; per-lane halo coordinates are computed once, per-tile addresses from them;
//   * the epilogue is conv_rows_epi.h's (skips, depth_to_space, paired 16-byte stores), issued after a tile's second chunk and never waited for.
// LDS: 72 KiB + 3 x 29.25 KiB + biases = 160 KiB exactly, one workgroup per CU, 12 waves, <= 168 registers.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "conv_rows_epi.h"

namespace {

using namespace convk;

__device__ __forceinline__ f32x4 mma16s(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int S_NCOMP = 8, S_NLOAD = 4, S_R = 3;
constexpr int S_TH = S_NCOMP * S_R, S_TW = 16, S_PH = S_TH + 2, S_PW = S_TW + 2, S_NPIX = S_PH * S_PW;      // 24 x 16 tile, 26 x 18 halo image
constexpr int S_BUF = S_NPIX * 64;                     // 29 952 B per unit
constexpr int S_NBUF = 3;
constexpr int S_WB = 2 * 9 * 4 * 1024;                 // both chunks' weights of one 64-cout tile
constexpr int S_LDS = S_WB + S_NBUF * S_BUF + 64 * 4;
constexpr int S_PIECES = (S_NPIX + 15) / 16;           // 30 DMA pieces of 16 pixels per unit (the last one: 4 pixels)
static_assert(S_LDS <= 160 * 1024, "LDS budget");

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct StreamParams {
    ConvParams c;                   // the conv as conv_rows sees it (tilesX / tilesY for 24 x 16 tiles)
    const char* zero;               // zero page (>= 64 B)
    int nct, tilesX, tilesY;        // cout tiles; spatial tiles per image
    int tiles_per_wg;               // a workgroup owns spatial tiles [wgs * tiles_per_wg, ...) of cout tile blockIdx.x % nct
    int ntiles;                     // B * tilesY * tilesX
};

__global__ void __launch_bounds__((S_NCOMP + S_NLOAD) * 64, (S_NCOMP + S_NLOAD + 3) / 4) conv64_stream_kernel(StreamParams sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lw = smem;
    char* const stg = smem + S_WB;
    float* const lbias = reinterpret_cast<float*>(stg + S_NBUF * S_BUF);
    constexpr int NTHR = (S_NCOMP + S_NLOAD) * 64;
    const ConvParams& p = sp.c;
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = blockIdx.x % sp.nct, wgs = blockIdx.x / sp.nct;
    const int t0 = wgs * sp.tiles_per_wg, t1 = min(sp.ntiles, t0 + sp.tiles_per_wg);
    if (t0 >= t1) return;                                              // whole workgroup
    const int nunits = 2 * (t1 - t0);
    const int H = p.H, W = p.W;

    // one-time: this cout tile's weights (row-sliding layout: [ct][chunk][tap][n][lane][8]) and biases
    {
        const char* wsrc = p.w + (int64_t)ct * S_WB;
        for (int u = tid; u < S_WB / 16; u += NTHR) *reinterpret_cast<f32x4*>(lw + u * 16) = *reinterpret_cast<const f32x4*>(wsrc + u * 16);
        if (tid < 64) lbias[tid] = p.bias[ct * 64 + tid];
    }
    auto tile_of = [&](int tt, int& b, int& y0, int& x0) {
        const int tx = tt % sp.tilesX;
        const int r = tt / sp.tilesX;
        const int ty = r % sp.tilesY;
        b = r / sp.tilesY;
        y0 = ty * S_TH; x0 = tx * S_TW;
    };
    if (wave >= S_NCOMP) {
        // ---------------------------------------------------------------------------------------------------- loader waves
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // own share of the weight copy
        const int lwr = wave - S_NCOMP;
        auto loader = [&](auto LWc) {
            constexpr int LW = decltype(LWc)::value;
            constexpr int NP = (S_PIECES - LW + 3) / 4;                  // pieces LW, LW + 4, ... of every unit: 8, 8, 7, 7
            // this lane's halo pixel in piece g: t = 16 g + lane / 4; slot lane % 4 holds slice (lane % 4) ^ 2 bit2(t)  (the row kernels' LDS image)
            int rel[NP];                                                 // (py << 16) | pxx, -1: beyond the image (last piece)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int t = 16 * (LW + 4 * i) + (lane >> 2);
                rel[i] = t < S_NPIX ? ((t / S_PW) << 16) | (t % S_PW) : -1;
            }
            const int slot = lane & 3;
            auto stage_unit = [&](int k) {
                const int c = k & 1;
                int b, y0, x0;
                tile_of(min(t0 + (k >> 1), t1 - 1), b, y0, x0);          // units past the end re-stage the last tile (nobody reads them)
                const char* const img = p.in + ((int64_t)b * H * W * p.in_cs + p.in_coff) * 2 + c * 64;
                char* const dst = stg + (k % S_NBUF) * S_BUF;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int t = 16 * (LW + 4 * i) + (lane >> 2);
                    const int gy = y0 + (rel[i] >> 16) - 1, gx = x0 + (rel[i] & 0xffff) - 1;
                    const bool inside = rel[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    const int sl = slot ^ ((t & 4) >> 1);
                    const char* src = inside ? img + ((int64_t)gy * W + gx) * (p.in_cs * 2) + sl * 16 : sp.zero + sl * 16;
                    if (rel[i] >= 0)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(dst + (LW + 4 * i) * 1024), 16, 0, 0);
                }
            };
            // the last piece of a unit is partial (pixels 464..467): the loader that owns it issues it with 16 active lanes, which still counts as one
            // vector-memory operation -- every loader's per-unit count NP is a constant
            stage_unit(0); stage_unit(1);
            for (int k = 0; k < nunits; ++k) {
                if (k == 0) wait_vm<NP>();                                // unit 0 has landed (unit 1 may fly)
                __builtin_amdgcn_s_barrier();                            // compute may start unit k; unit k - 1's buffer is free
                stage_unit(k + 2);
                wait_vm<NP>();                                            // unit k + 1 has landed
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (lwr == 0) loader(std::integral_constant<int, 0>{});
        else if (lwr == 1) loader(std::integral_constant<int, 1>{});
        else if (lwr == 2) loader(std::integral_constant<int, 2>{});
        else loader(std::integral_constant<int, 3>{});
        return;
    }
    // -------------------------------------------------------------------------------------------------------- compute waves
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int tbase = wave * S_R * S_PW + px;
    auto xaddr = [&](int yi, int kx) {
        const int tt = tbase + yi * S_PW + kx;
        return tt * 64 + ((q * 16) ^ ((tt & 4) << 3));
    };
    // the tile loop is instantiated per epilogue kind (conv_rows_epi.h), so that only that variant's loop invariants stay live across the MFMA loops
    auto tiles = [&](auto KINDc) {
    constexpr int KIND = decltype(KINDc)::value;
    for (int ti = t0; ti < t1; ++ti) {
        f32x4 acc[S_R][4];
#pragma unroll
        for (int r = 0; r < S_R; ++r)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = 2 * (ti - t0) + c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* const lin = stg + (k % S_NBUF) * S_BUF;
            const char* const lwc = lw + c * (9 * 4 * 1024);
            bf16x8 xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(0, 0));
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8 wf[3][4];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int n = 0; n < 4; ++n) wf[ky][n] = *reinterpret_cast<const bf16x8*>(lwc + ((ky * 3 + kx) * 4 + n) * 1024 + lane * 16);
#pragma unroll
                for (int yi = 0; yi < S_R + 2; ++yi) {
                    const bf16x8 xf = xcur;
                    if (yi + 1 < S_R + 2) xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(yi + 1, kx));
                    else if (kx + 1 < 3) xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(0, kx + 1));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int yo = yi - ky;
                        if (yo >= 0 && yo < S_R) {
#pragma unroll
                            for (int n = 0; n < 4; ++n) acc[yo][n] = mma16s(wf[ky][n], xf, acc[yo][n]);
                        }
                    }
                }
            }
        }
        int b, y0, x0;
        tile_of(ti, b, y0, x0);
        f32x4 biasv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) biasv[n] = *reinterpret_cast<const f32x4*>(lbias + n * 16 + 4 * q);
        rows_epilogue_as<4, S_R, KIND>(p, acc, biasv, b, y0, x0, ct, wave, px, q);
    }
    };
    switch (rows_epilogue_kind<4>(p)) {                                  // (the host sends only these kinds here: conv_stream_supported)
        case 0: tiles(std::integral_constant<int, 0>{}); break;
        case 1: tiles(std::integral_constant<int, 1>{}); break;
        case 2: tiles(std::integral_constant<int, 2>{}); break;
        default: tiles(std::integral_constant<int, 3>{}); break;
    }
}

}  // namespace

bool conv_stream_supported(const ConvWeights& w, const convk::ConvParams& p) {
    static const bool off = getenv("SR355_NO_CONV_STREAM") != nullptr;
    return !off && w.rows && w.dtype == SR_DTYPE_BF16 && w.NT == 4 && w.nchunks == 2 && w.Cin == 64 && w.CoutP == w.Cout && p.in_ps == 32 && p.in_cs >= 64 && p.in_cs % 8 == 0 &&
           p.in_coff % 8 == 0 && p.H % S_TH == 0 && p.W % S_TW == 0 && p.skip_lds == 0 && !p.f2w && !p.pjw && !p.plout && !p.dbg && p.cell_h == 0 && p.cell_w == 0 &&
           // the vector epilogues with at most one skip (conv_rows_epi.h kinds 0..3; the host passes a lone skip as skip 1)
           p.vec != 0 && (p.Cout & 3) == 0 && p.act != SR_ACT_TANH && (p.r <= 1 || (p.Cd & 15) == 0) && !p.s2;
}

int conv_stream_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p0, hipStream_t st) {
    if (!ctx->zero_page) return ctx->fail(SR_ERR_STATE, "context has no zero page");      // sr_init allocates and clears it
    StreamParams sp;
    sp.c = p0;
    sp.zero = static_cast<const char*>(ctx->zero_page);
    sp.nct = w.CoutP / 64;
    sp.tilesX = p0.W / S_TW; sp.tilesY = p0.H / S_TH;
    sp.c.tilesX = sp.tilesX; sp.c.tilesY = sp.tilesY;
    const int64_t ntiles = (int64_t)p0.B * sp.tilesX * sp.tilesY;
    if (ntiles >= (1ll << 30)) return ctx->fail(SR_ERR_INVALID, "conv_stream: too many tiles for one launch");
    sp.ntiles = (int)ntiles;
    const int ncu = ctx->cu_count();
    const int groups = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, ncu / sp.nct > 0 ? ncu / sp.nct : 1));      // workgroups per cout tile
    sp.tiles_per_wg = (int)((ntiles + groups - 1) / groups);
    const int ngroups = (int)((ntiles + sp.tiles_per_wg - 1) / sp.tiles_per_wg);
    auto k = conv64_stream_kernel;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), S_LDS)) return rc;
    hipLaunchKernelGGL(k, dim3(ngroups * sp.nct), dim3((S_NCOMP + S_NLOAD) * 64), S_LDS, st, sp);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}
