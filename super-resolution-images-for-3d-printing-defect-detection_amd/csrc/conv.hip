// conv.hip -- NHWC stride-1 SAME convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces the cuDNN kernels TensorFlow picks for keras Conv2D(padding="same") on the reference's
// hot path (SRCNN_model.py:50-52, EDSR_model.py:61-121, ESRGAN_model.py:230-341, VGG16 base).
//
// Formulation.  D^T[cout][pixel] += W^T[cout][k] * X^T[k][pixel] with v_mfma_f32_32x32x16_bf16
// (bf16) or 4 x v_mfma_f32_32x32x2_f32 (fp32, exact fp32 fma chain).  The weight fragment is
// the MFMA "A" operand and the pixel fragment the "B" operand, so every lane ends up owning one
// output pixel and groups of 4 consecutive output channels -> 8/16-byte NHWC stores.
//
// Work split.  A workgroup (4 waves) owns a TH x 16 pixel tile (TH = 8*MT) of one image and
// NT*32 output channels.  Wave w owns rows [2*MT*w, 2*MT*(w+1)); an "M-block" is 2 rows x 16
// columns = 32 pixels = the 32 columns of one MFMA.
//
// LDS.  The input halo tile (TH+KS-1) x (16+KS-1) pixels of one Cin chunk is staged pixel-major
// with CS = chunk_bytes + 16 bytes per pixel (odd number of 16-byte slots -> the 16 lanes a
// ds_read_b128 services together land on 16 distinct slots).  A tap (ky,kx) is then a constant
// byte offset from a lane's base address.  Weights are pre-packed on the host in exactly the
// order the lanes read them (1 KiB per (tap, k-group, cout-block), lane-linear), so the weight
// stage is a straight copy and its ds_read_b128 is conflict free.
//
// k-group = the 16 bytes lane-half 0 and the 16 bytes lane-half 1 feed to one MFMA step:
//   wide (Cin*sizeof >= 64 B): both halves sit on the same tap, consecutive channel slices;
//   thin (Cin fits one 16-byte slice, e.g. RGB padded to 8 bf16 / 4 fp32): half 0 takes tap 2g,
//   half 1 tap 2g+1.
//
// HBM traffic per launch (algorithmic): B*H*W*(Cin + Cout)*sizeof(T) (+ skips); FLOP
// 2*B*H*W*KS^2*Cin*Cout.  DESIGN.md prices each layer against both roofs.
#include <type_traits>

#include "conv_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

using namespace convk;

template <typename T> struct TT;
template <> struct TT<bf16_t> { static constexpr int E = 8; typedef bf16x8 frag; };
template <> struct TT<float>  { static constexpr int E = 4; typedef f32x4 frag; };

__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma(f32x4 a, f32x4 b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
// the first NV of a 16-byte slice's channels only: an RGB image padded to four fp32 channels carries a zero in the fourth (weights and
// pixels alike), whose MFMA adds exactly 0 -- skipping it is the same sum with a quarter fewer matrix instructions (round 4)
template <int NV> __device__ __forceinline__ f32x16 mma_nv(f32x4 a, f32x4 b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < NV; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
template <int NV> __device__ __forceinline__ f32x16 mma_nv(bf16x8 a, bf16x8 b, f32x16 c) { return mma(a, b, c); }

// Epilogue of the 32x32 kernels: lane (r,h) owns pixel r of each M-block and couts 8q+4h..+3.
template <typename T, int NT, int MT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[MT][NT], int b, int y0, int x0, int ct,
                                              int wave, int r, int h) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int oy = y0 + (wave * MT + m) * 2 + (r >> 4);
        const int ox = x0 + (r & 15);
        if (oy >= p.H || ox >= p.W) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a[4] = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
                epilogue4<T>(p, b, oy, ox, (ct * NT + n) * 32 + 8 * q + 4 * h, a);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wide kernel: Cin*sizeof(T) is a multiple of KGPT*32 bytes
// ------------------------------------------------------------------------------------------------
template <typename T, int KS, int KGPT, int NT, int MT>
__global__ void __launch_bounds__(256) conv_wide_kernel(ConvParams p) {
    typedef typename TT<T>::frag frag;
    constexpr int E = TT<T>::E;
    constexpr int TH = 8 * MT, TW = 16;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PADK = (KS - 1) / 2, NTAP = KS * KS;
    constexpr int SPP = KGPT * 2;
    constexpr int CS = KGPT * 32 + 16;
    constexpr int NIN = PH * PW * SPP;
    constexpr int NINT = (NIN + 255) / 256;
    constexpr int WUNITS = NTAP * KGPT * NT * 64;
    constexpr int LIN_BYTES = (PH * PW * CS + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lin = smem;
    char* lw = smem + LIN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % p.tilesX; t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int b = t / p.tilesY;
    const int ct = blockIdx.y;
    const int y0 = ty * TH, x0 = tx * TW;
    const T* inb = reinterpret_cast<const T*>(p.in) + (int64_t)b * p.H * p.W * p.in_cs + p.in_coff;

    // staging descriptors: unit u -> (halo pixel, 16-byte slice)
    int soff[NINT];   // element offset of the slice inside the image (chunk 0); -1: outside the image (zero)
    int doff[NINT];   // LDS byte offset; -1: no unit
#pragma unroll
    for (int i = 0; i < NINT; ++i) {
        const int u = tid + 256 * i;
        const int pix = u / SPP, sl = u - pix * SPP;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - PADK, gx = x0 + px - PADK;
        const bool live = u < NIN;
        const bool inside = live && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        doff[i] = live ? pix * CS + sl * 16 : -1;
        soff[i] = inside ? (int)(((int64_t)gy * p.W + gx) * p.in_cs) + sl * E : -1;
    }
    frag pre[NINT];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NINT; ++i) {
            const int so = soff[i];
            frag v = *reinterpret_cast<const frag*>(inb + (so >= 0 ? so + chunk * (SPP * E) : 0));   // always a valid address
            frag z = {};
            pre[i] = so >= 0 ? v : z;
        }
    };

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = (((wave * MT + m) * 2 + (r >> 4)) * PW + (r & 15)) * CS + h * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    issue(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        __syncthreads();   // previous chunk's MFMAs have read their operands
#pragma unroll
        for (int i = 0; i < NINT; ++i)
            if (doff[i] >= 0) *reinterpret_cast<frag*>(lin + doff[i]) = pre[i];
        {
            const char* wsrc = p.w + ((int64_t)ct * p.nchunks + chunk) * (int64_t)(WUNITS * 16);
            for (int u = tid; u < WUNITS; u += 256)
                *reinterpret_cast<f32x4*>(lw + u * 16) = *reinterpret_cast<const f32x4*>(wsrc + u * 16);
        }
        __syncthreads();
        if (chunk + 1 < p.nchunks) issue(chunk + 1);   // next chunk's global loads fly under the MFMAs
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int toff = ((tap / KS) * PW + (tap % KS)) * CS;
#pragma unroll
            for (int kg = 0; kg < KGPT; ++kg) {
                frag wf[NT], xf[MT];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    wf[n] = *reinterpret_cast<const frag*>(lw + ((tap * KGPT + kg) * NT + n) * 1024 + lane * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m) xf[m] = *reinterpret_cast<const frag*>(lin + abase[m] + toff + kg * 32);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = mma(wf[n], xf[m], acc[m][n]);
            }
        }
    }
    conv_epilogue<T, NT, MT>(p, acc, b, y0, x0, ct, wave, r, h);
}

// ------------------------------------------------------------------------------------------------
// wide kernel, split-K inside the workgroup (fp32, 3x3; round 4).  The training step's layers are SMALL: 16 patches of 24 x 24 are 9216 pixels,
// 72 of the 8 x 16 tiles above on 256 CUs, and every one of their waves walks the whole K = 9 * Cin alone (conv5 of a dense block: 1728 fp32
// MFMAs of 64 cycles, 67 us for 2 GFLOP = 20 % of the fp32 peak, profiles/r03_cfg3_train_kernel_stats.csv).  Here a workgroup owns ONE M-block
// (2 rows x 16 columns = the 32 columns of one MFMA) and its four waves share the k-groups of every staged chunk (wave w takes k-groups w, w + 4, ...);
// at the end the four partial accumulators meet in LDS and wave w adds them -- always in the order 0, 1, 2, 3: deterministic, no atomics -- for the
// four-cout groups q = w, then runs the usual epilogue.  Four times the workgroups, a quarter of the serial MFMA chain each; the weights of a chunk
// are staged per 32 pixels instead of per 128 (L2 -> LDS at 59 B/clk per CU: ~3.5 us for conv5's 432 KiB, under the 13 us of MFMAs).
// (Round 4 also measured the chunk's weights by LDS-DMA into two buffers, a chunk ahead, instead of through registers: 55.8 ms per cfg3 step against 54.5-56.7 --
//  the weight stage is not what a chunk waits for; removed.)
// ------------------------------------------------------------------------------------------------
template <int KS, int KGPT, int NT>
__global__ void __launch_bounds__(256) conv_wide_sk_kernel(ConvParams p) {
    typedef f32x4 frag;
    constexpr int E = 4;
    constexpr int TH = 2, TW = 16;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PADK = (KS - 1) / 2, NTAP = KS * KS;
    constexpr int SPP = KGPT * 2;
    constexpr int CS = KGPT * 32 + 16;
    constexpr int NIN = PH * PW * SPP;
    constexpr int NINT = (NIN + 255) / 256;
    constexpr int NKG = NTAP * KGPT;
    constexpr int WUNITS = NKG * NT * 64;
    constexpr int LIN_BYTES = (PH * PW * CS + 15) & ~15;
    static_assert(WUNITS * 16 >= 4 * NT * 16 * 64 * 4, "the reduction reuses the weight stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lin = smem;
    char* lw = smem + LIN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % p.tilesX; t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int b = t / p.tilesY;
    const int ct = blockIdx.y;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* inb = reinterpret_cast<const float*>(p.in) + (int64_t)b * p.H * p.W * p.in_cs + p.in_coff;

    int soff[NINT], doff[NINT];
#pragma unroll
    for (int i = 0; i < NINT; ++i) {
        const int u = tid + 256 * i;
        const int pix = u / SPP, sl = u - pix * SPP;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - PADK, gx = x0 + px - PADK;
        const bool live = u < NIN;
        const bool inside = live && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        doff[i] = live ? pix * CS + sl * 16 : -1;
        soff[i] = inside ? (int)(((int64_t)gy * p.W + gx) * p.in_cs) + sl * E : -1;
    }
    // both the next chunk's pixels and its weights fly under the current chunk's MFMAs: with a quarter of the chain per wave the weight stage's
    // round trip (36 KiB per chunk at NT = 2) would otherwise be as long as the chunk's arithmetic
    constexpr int NWT = (WUNITS + 255) / 256;
    frag pre[NINT], wpre[NWT];
    auto issue = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NINT; ++i) {
            const int so = soff[i];
            frag v = *reinterpret_cast<const frag*>(inb + (so >= 0 ? so + chunk * (SPP * E) : 0));
            frag z = {};
            pre[i] = so >= 0 ? v : z;
        }
        const char* wsrc = p.w + ((int64_t)ct * p.nchunks + chunk) * (int64_t)(WUNITS * 16);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int u = tid + 256 * i;
            wpre[i] = *reinterpret_cast<const f32x4*>(wsrc + (u < WUNITS ? u : 0) * 16);
        }
    };
    const int abase = ((r >> 4) * PW + (r & 15)) * CS + h * 16;
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

    issue(0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NINT; ++i)
            if (doff[i] >= 0) *reinterpret_cast<frag*>(lin + doff[i]) = pre[i];
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int u = tid + 256 * i;
            if (u < WUNITS) *reinterpret_cast<f32x4*>(lw + u * 16) = wpre[i];
        }
        __syncthreads();
        if (chunk + 1 < p.nchunks) issue(chunk + 1);
#pragma unroll
        for (int g0 = 0; g0 < NKG; g0 += 4) {
            const int g = g0 + wave;                               // wave-uniform
            if (g < NKG) {
                const int tap = g / KGPT, kg = g - tap * KGPT;
                const int toff = ((tap / KS) * PW + (tap % KS)) * CS;
                const frag xf = *reinterpret_cast<const frag*>(lin + abase + toff + kg * 32);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const frag wf = *reinterpret_cast<const frag*>(lw + (g * NT + n) * 1024 + lane * 16);
                    acc[n] = mma(wf, xf, acc[n]);
                }
            }
        }
    }
    // the four partial sums of every output meet in LDS (the weight stage is free now) and are added in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lw);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wave * NT + n) * 16 + i) * 64 + lane] = acc[n][i];
    __syncthreads();
    const int oy = y0 + (r >> 4), ox = x0 + (r & 15);
    if (oy >= p.H || ox >= p.W) return;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * wave + e;
            float v = red[((0 * NT + n) * 16 + i) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) v += red[((w2 * NT + n) * 16 + i) * 64 + lane];
            a[e] = v;
        }
        epilogue4<float>(p, b, oy, ox, (ct * NT + n) * 32 + 8 * wave + 4 * h, a);
    }
}

template <int KS, int KGPT, int NT>
int launch_wide_sk(sr_ctx* ctx, const ConvParams& p0, int nct, hipStream_t st) {
    constexpr int PH = 2 + KS - 1, PW = 16 + KS - 1;
    constexpr int CS = KGPT * 32 + 16;
    constexpr int lds = ((PH * PW * CS + 15) & ~15) + KS * KS * KGPT * NT * 1024;
    ConvParams p = p0;
    p.tilesX = (p.W + 15) / 16;
    p.tilesY = (p.H + 1) / 2;
    auto kern = conv_wide_sk_kernel<KS, KGPT, NT>;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    dim3 grid((unsigned)((int64_t)p.tilesX * p.tilesY * p.B), (unsigned)nct);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// thin kernel: one 16-byte slice per pixel holds every input channel (RGB padded to 8 bf16 / 4 fp32)
// ------------------------------------------------------------------------------------------------
constexpr int THIN_GPS = 16;   // k-groups per weight stage

// NV: fp32 only -- valid channels of the slice (3 for an RGB image: the padding channel's MFMA is skipped).
// PW2: fp32 only -- a 1x1 conv to <= 32 channels (p.pw2w, Pw2Weights) is computed from the accumulators in the epilogue and stored INSTEAD of this conv's
// output: the workgroup owns all NT * 32 couts of its pixels (gridDim.y == 1).  In the 32x32 accumulator layout lane (r, h) holds, for pixel r of an M-block,
// couts 32 n + 8 (i / 4) + 4 h + (i % 4) in element i of block n -- which IS the B operand of a 32x32x2 fp32 MFMA whose two k values are the couts lane
// halves 0 and 1 hold in the same element: the second conv is NT * 16 MFMAs per M-block on the activated accumulators, no cross-lane movement.
// GPS: k-groups per weight stage (the fused variant: 8 x 16-pixel tiles, 8 k-groups per stage: 30 KiB of LDS and ~110 registers, four workgroups per CU).
template <typename T, int KS, int NT, int MT, int NV = 4, bool PW2 = false, int GPS = THIN_GPS>
__global__ void __launch_bounds__(256) conv_thin_kernel(ConvParams p) {
    typedef typename TT<T>::frag frag;
    constexpr int TH = 8 * MT, TW = 16;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1, PADK = (KS - 1) / 2, NTAP = KS * KS;
    constexpr int KGT = (NTAP + 1) / 2;
    constexpr int NIN = PH * PW;
    constexpr int LIN_BYTES = NIN * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lin = smem;
    char* lw = smem + LIN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % p.tilesX; t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int b = t / p.tilesY;
    const int ct = blockIdx.y;
    const int y0 = ty * TH, x0 = tx * TW;
    const T* inb = reinterpret_cast<const T*>(p.in) + (int64_t)b * p.H * p.W * p.in_cs + p.in_coff;

    for (int u = tid; u < NIN; u += 256) {
        const int py = u / PW, px = u - py * PW;
        const int gy = y0 + py - PADK, gx = x0 + px - PADK;
        const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        frag v = *reinterpret_cast<const frag*>(inb + (inside ? ((int64_t)gy * p.W + gx) * p.in_cs : 0));
        frag z = {};
        *reinterpret_cast<frag*>(lin + u * 16) = inside ? v : z;
    }
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = (((wave * MT + m) * 2 + (r >> 4)) * PW + (r & 15)) * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    for (int g0 = 0; g0 < KGT; g0 += GPS) {
        const int ng = min(GPS, KGT - g0);
        __syncthreads();
        if constexpr (PW2) {
            // the head's weights are packed for one 32-cout block per workgroup, [cout block][k-group][lane]: this workgroup takes all NT blocks
            for (int u = tid; u < ng * NT * 64; u += 256) {
                const int l = u & 63, n = (u >> 6) % NT, gi = (u >> 6) / NT;
                *reinterpret_cast<f32x4*>(lw + u * 16) = *reinterpret_cast<const f32x4*>(p.w + (((int64_t)n * KGT + g0 + gi) * 64 + l) * 16);
            }
        } else {
            const char* wsrc = p.w + ((int64_t)ct * KGT + g0) * (int64_t)(NT * 1024);
            for (int u = tid; u < ng * NT * 64; u += 256)
                *reinterpret_cast<f32x4*>(lw + u * 16) = *reinterpret_cast<const f32x4*>(wsrc + u * 16);
        }
        __syncthreads();
#pragma unroll 2
        for (int gi = 0; gi < ng; ++gi) {
            int tap = 2 * (g0 + gi) + h;
            tap = tap < NTAP ? tap : 0;   // the odd leftover slot carries zero weights
            const int toff = ((tap / KS) * PW + (tap % KS)) * 16;
            frag wf[NT], xf[MT];
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const frag*>(lw + (gi * NT + n) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m) xf[m] = *reinterpret_cast<const frag*>(lin + abase[m] + toff);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = mma_nv<NV>(wf[n], xf[m], acc[m][n]);
        }
    }
    if constexpr (PW2 && std::is_same<T, float>::value) {
        __syncthreads();                                           // every wave is done with the weight stage: the 1x1's A operands take its place
        float* l2 = reinterpret_cast<float*>(lw);
        for (int u = tid; u < NT * 16 * 64; u += 256) l2[u] = p.pw2w[u];
        __syncthreads();
        ConvParams q = p;
        q.bias = p.pw2bias; q.Cout = p.pw2_cout; q.act = p.pw2_act; q.alpha = 1.f;
        f32x16 acc2[MT][1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[m][0][e] = 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = 32 * n + 8 * (i >> 2) + 4 * h + (i & 3);
                    const float v = act_apply(acc[m][n][i] + p.bias[c], p.act) * p.alpha;
                    acc2[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(l2[(n * 16 + i) * 64 + lane], v, acc2[m][0], 0, 0, 0);
                }
        }
        conv_epilogue<T, 1, MT>(q, acc2, b, y0, x0, 0, wave, r, h);
    } else {
        conv_epilogue<T, NT, MT>(p, acc, b, y0, x0, ct, wave, r, h);
    }
}

constexpr int MT_DEFAULT = 3;   // 24 x 16 pixel tiles: 48/96/192-pixel patches tile exactly

template <typename T, int KS, int KGPT, int NT, int MT>
int launch_wide_mt(sr_ctx* ctx, const ConvParams& p0, int nct, hipStream_t st) {
    constexpr int PH = 8 * MT + KS - 1, PW = 16 + KS - 1, CS = KGPT * 32 + 16;
    constexpr int lds = ((PH * PW * CS + 15) & ~15) + KS * KS * KGPT * NT * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    ConvParams p = p0;
    p.tilesX = (p.W + 15) / 16;
    p.tilesY = (p.H + 8 * MT - 1) / (8 * MT);
    auto kern = conv_wide_kernel<T, KS, KGPT, NT, MT>;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    dim3 grid((unsigned)((int64_t)p.tilesX * p.tilesY * p.B), (unsigned)nct);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

template <typename T, int KS, int KGPT, int NT>
int launch_wide(sr_ctx* ctx, const ConvParams& p0, int nct, hipStream_t st) {
    // Small fp32 problems (the training step's 16 x 24 x 24 batches: 9216 pixels) give the 24 x 16 tiling 32 workgroups per cout tile on
    // 256 CUs -- a launch then takes as long as one workgroup's whole K loop (round 2: 146 us for 0.85 GFLOP).  8 x 16 tiles give three
    // times the workgroups and a third of the serial work each.  (bf16 inference never gets here with so few pixels.)
    if constexpr (std::is_same<T, float>::value) {
        const int64_t wgs3 = (int64_t)((p0.W + 15) / 16) * ((p0.H + 8 * MT_DEFAULT - 1) / (8 * MT_DEFAULT)) * p0.B * nct;
        if constexpr (KS == 3 && KGPT == 2) {
            // still fewer than one 8 x 16 tile per CU: split K inside the workgroup (conv_wide_sk_kernel) -- 4 x the workgroups, a quarter of the chain each
            static const bool no_sk = getenv("SR355_NO_SPLITK") != nullptr;      // A/B switch (diagnostic)
            const int64_t wgs1 = (int64_t)((p0.W + 15) / 16) * ((p0.H + 7) / 8) * p0.B * nct;
            if (!no_sk && p0.splitk_ok && wgs1 < ctx->cu_count()) return launch_wide_sk<KS, KGPT, NT>(ctx, p0, nct, st);
        }
        if (wgs3 < 2 * ctx->cu_count()) return launch_wide_mt<T, KS, KGPT, NT, 1>(ctx, p0, nct, st);
    }
    return launch_wide_mt<T, KS, KGPT, NT, MT_DEFAULT>(ctx, p0, nct, st);
}

template <typename T, int KS, int NT, int NV = 4, bool PW2 = false, int MT = MT_DEFAULT, int GPS = THIN_GPS>
int launch_thin(sr_ctx* ctx, const ConvParams& p0, int nct, hipStream_t st) {
    constexpr int PH = 8 * MT + KS - 1, PW = 16 + KS - 1;
    constexpr int KGT = (KS * KS + 1) / 2;
    constexpr int lds = PH * PW * 16 + (KGT < GPS ? KGT : GPS) * NT * 1024;
    static_assert(!PW2 || (KGT < GPS ? KGT : GPS) * NT * 1024 >= NT * 16 * 64 * 4, "the fused 1x1's operands fit the weight stage");
    ConvParams p = p0;
    p.tilesX = (p.W + 15) / 16;
    p.tilesY = (p.H + 8 * MT - 1) / (8 * MT);
    auto kern = conv_thin_kernel<T, KS, NT, MT, NV, PW2, GPS>;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    dim3 grid((unsigned)((int64_t)p.tilesX * p.tilesY * p.B), (unsigned)nct);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

template <typename T, int KS, int KGPT>
int dispatch_wide_nt(sr_ctx* ctx, const ConvParams& p, int NT, int nct, hipStream_t st) {
    switch (NT) {
        case 1: return launch_wide<T, KS, KGPT, 1>(ctx, p, nct, st);
        case 2: return launch_wide<T, KS, KGPT, 2>(ctx, p, nct, st);
        case 3: return launch_wide<T, KS, KGPT, 3>(ctx, p, nct, st);
    }
    return ctx->fail(SR_ERR_INVALID, "conv: unsupported NT");
}
template <typename T, int KS>
int dispatch_thin_nt(sr_ctx* ctx, const ConvParams& p, int NT, int nct, hipStream_t st) {
    switch (NT) {
        case 1: return launch_thin<T, KS, 1>(ctx, p, nct, st);
        case 2: return launch_thin<T, KS, 2>(ctx, p, nct, st);
        case 3: return launch_thin<T, KS, 3>(ctx, p, nct, st);
    }
    return ctx->fail(SR_ERR_INVALID, "conv: unsupported NT");
}

template <typename T>
int dispatch(sr_ctx* ctx, const ConvWeights& w, const ConvParams& p, int nct, hipStream_t st) {
    if constexpr (std::is_same<T, float>::value) {
        // SRCNN's head (SRCNN_model.py:50): 9x9 on an RGB image to 96 channels, optionally with the 1x1 that follows it in the epilogue
        // (the packed weights of a thin conv are [cout tile of 32][k-group][lane][4]: NT = 1 with three cout tiles per pixel tile when the head runs alone --
        // three waves per SIMD, which beat one wave per SIMD at NT = 3 by 1.8x in round 1 --; the fused variant needs all 96 couts of a pixel in one
        // workgroup and reads the same bytes as [k-group][cout tile] through a stride: see the weight stage)
        if (w.thin && w.KS == 9 && w.Cin == 3 && w.CoutP == 96) {
            // tile height / weight-stage depth measured on 4 x 1024 x 1024 images (round 4, same box): 8 x 16 tiles with 8 k-groups per stage 2.04 ms,
            // 4 per stage 2.09, 16 x 16 tiles 2.32-2.35, 24 x 16 tiles (one wave per SIMD) 3.05 -- occupancy, not weight traffic, is what this kernel wants
            if (p.pw2w) return launch_thin<T, 9, 3, 3, true, 1, 8>(ctx, p, 1, st);
            return launch_thin<T, 9, 1, 3, false>(ctx, p, nct, st);
        }
    }
    if (p.pw2w) return ctx->fail(SR_ERR_INVALID, "conv: the fused 1x1 follows the fp32 9x9 RGB head only");
    if (w.thin) {
        switch (w.KS) {
            case 3: return dispatch_thin_nt<T, 3>(ctx, p, w.NT, nct, st);
            case 5: return dispatch_thin_nt<T, 5>(ctx, p, w.NT, nct, st);      // input gradient of SRCNN's 5x5 32 -> 3 conv (3 -> 32 on dy)
            case 9: return dispatch_thin_nt<T, 9>(ctx, p, w.NT, nct, st);
        }
    } else {
        if (w.KS == 1 && w.KGPT == 4) return dispatch_wide_nt<T, 1, 4>(ctx, p, w.NT, nct, st);
        if (w.KS == 1 && w.KGPT == 2) return dispatch_wide_nt<T, 1, 2>(ctx, p, w.NT, nct, st);
        if (w.KS == 3 && w.KGPT == 2) return dispatch_wide_nt<T, 3, 2>(ctx, p, w.NT, nct, st);
        if (w.KS == 5 && w.KGPT == 2 && w.NT == 1) return launch_wide<T, 5, 2, 1>(ctx, p, nct, st);
    }
    return ctx->fail(SR_ERR_INVALID, "conv: unsupported kernel size " + std::to_string(w.KS));
}

inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                          // round to nearest even
    return (uint16_t)(u >> 16);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// fp32 convs with at most 4 output channels (SRCNN's 5x5 32->3 tail, the fp32 models' RGB tails).  On the 32-cout MFMA tile
// these compute 32 couts to keep 3 (1.9 ms of SRCNN's 4.9 ms); here a thread owns one output pixel and its <= 4 couts on the
// VALU: the halo tile of 4 input channels at a time sits in LDS (16 B per pixel, conflict-free ds_read_b128), the weights are
// wave-uniform and come through the scalar cache ([tap][cin][4] floats), one v_fma per (tap, cin, cout) in a fixed order.
// ---------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256) conv_fewcout_f32_kernel(ConvParams p) {
    constexpr int TS = 16, PS = TS + KS - 1, PADK = (KS - 1) / 2;
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z, y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
    const int H = p.H, W = p.W;
    const float* inb = reinterpret_cast<const float*>(p.in) + (int64_t)b * H * W * p.in_cs + p.in_coff;
    const f32x4* __restrict__ wk = reinterpret_cast<const f32x4*>(p.w);
    const int CinP = p.nchunks * 4;
    // (round 4: the four couts as two v_pk_fma_f32 pairs -- pixel value broadcast by op_sel, weight pair in SGPRs, half the vector instructions -- measured
    //  0.78 ms against 0.73 for 4 x 1024 x 1024; the same loop on v_mfma_f32_4x4x1 (64 pixels x 4 couts per instruction, bit-identical sums) 0.76: the kernel
    //  is not bound by its arithmetic issue either way; both reverted)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = 0; c4 < p.nchunks; ++c4) {
        __syncthreads();
        for (int u = tid; u < PS * PS; u += 256) {
            const int py = u / PS, px = u - py * PS;
            const int gy = y0 + py - PADK, gx = x0 + px - PADK;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *reinterpret_cast<const f32x4*>(inb + ((int64_t)gy * W + gx) * p.in_cs + c4 * 4);
            *reinterpret_cast<f32x4*>(tile + u * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(tile + ((ty + ky) * PS + tx + kx) * 4);
                const f32x4* wt = wk + (size_t)(ky * KS + kx) * CinP + c4 * 4;      // wave-uniform
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const f32x4 wv = wt[ci];
#pragma unroll
                    for (int co = 0; co < 4; ++co) acc[co] = fmaf(xv[ci], wv[co], acc[co]);
                }
            }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < H && ox < W) epilogue4<float>(p, b, oy, ox, 0, acc);
}

// The same conv with the WHOLE channel depth of the halo tile staged at once (round 4).  The kernel above stages four channels at a time: 16 bytes out of every
// pixel's 128-byte line per pass, eight passes -- the cache hierarchy moves every line eight times (6.7 GB for SRCNN's 5x5 tail on 4 x 1024 x 1024, which is what
// its 0.73 ms were; neither packed FMAs nor the matrix core moved it).  Here a pixel's CinP channels are staged once, as whole lines, at a pixel stride of
// CinP * 4 + 16 bytes (an odd number of 16-byte slots: the 16 lanes of a ds_read_b128 group land on distinct slots), all the layer's weights beside them as
// [chunk][tap][cout][ci], and the multiply-adds run on v_mfma_f32_4x4x1_16B_f32: sixteen independent 4 x 4 x 1 blocks per instruction = 4 couts x 64 pixels, a lane's
// four result registers its pixel's couts.  One k step = one (tap, channel) in the order (4-channel chunk, ky, kx, channel) of the kernel above, each a fused
// multiply-add per output.
template <int KS>
__global__ void __launch_bounds__(256) conv_fewcout_full_kernel(ConvParams p) {
    constexpr int TS = 16, PS = TS + KS - 1, PADK = (KS - 1) / 2, NTAP = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CinP = p.nchunks * 4, PSTR = CinP + 4;                        // floats per staged pixel
    float* tile = reinterpret_cast<float*>(smem);
    float* wl = tile + PS * PS * PSTR;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z, y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
    const int H = p.H, W = p.W;
    const float* inb = reinterpret_cast<const float*>(p.in) + (int64_t)b * H * W * p.in_cs + p.in_coff;
    const float* __restrict__ wk = reinterpret_cast<const float*>(p.w);      // [tap][CinP][4]
    const int nsl = p.nchunks;                                               // 16-byte slices per pixel
    for (int u = tid; u < PS * PS * nsl; u += 256) {
        const int pix = u / nsl, sl = u - pix * nsl;
        const int py = pix / PS, px = pix - py * PS;
        const int gy = y0 + py - PADK, gx = x0 + px - PADK;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *reinterpret_cast<const f32x4*>(inb + ((int64_t)gy * W + gx) * p.in_cs + sl * 4);
        *reinterpret_cast<f32x4*>(tile + pix * PSTR + sl * 4) = v;
    }
    for (int u = tid; u < nsl * NTAP * 16; u += 256) {                        // wl[c4][tap][co][ci] <- wk[tap][c4 * 4 + ci][co]
        const int ci = u & 3, co = (u >> 2) & 3, tap = (u >> 4) % NTAP, c4 = (u >> 4) / NTAP;
        wl[u] = wk[((size_t)tap * CinP + c4 * 4 + ci) * 4 + co];
    }
    __syncthreads();
    // four independent chains, one per channel of a slice (a single chain of 800 dependent MFMAs ran at their latency with two waves per SIMD: 0.60 ms for
    // SRCNN's tail on 4 x 1024 x 1024), added pairwise at the end
    f32x4 acc[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) acc[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c4 = 0; c4 < nsl; ++c4) {
        const float* wc = wl + (c4 * NTAP * 4 + (tid & 3)) * 4;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(tile + ((ty + ky) * PS + tx + kx) * PSTR + c4 * 4);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wc + (ky * KS + kx) * 16);
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) acc[ci] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[ci], xv[ci], acc[ci], 0, 0, 0);
            }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    float a4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    if (oy < H && ox < W) epilogue4<float>(p, b, oy, ox, 0, a4);
}

template <int KS>
int launch_fewcout(sr_ctx* ctx, const ConvParams& p, hipStream_t st) {
    if (p.B > 65535 || (p.H + 15) / 16 > 65535) return ctx->fail(SR_ERR_INVALID, "conv: too many tiles for one launch");
    constexpr int PS = 16 + KS - 1;
    const int CinP = p.nchunks * 4;
    const int lds = PS * PS * (CinP + 4) * 4 + p.nchunks * KS * KS * 16 * 4;
    static const bool chunked = getenv("SR355_FEW_CHUNKED") != nullptr;      // A/B switch (diagnostic): round 3's four-channels-at-a-time kernel
    if (!chunked && lds <= 80 * 1024 && (p.in_cs & 3) == 0 && (p.in_coff & 3) == 0) {      // two workgroups per CU
        auto kern = conv_fewcout_full_kernel<KS>;
        if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3((p.W + 15) / 16, (p.H + 15) / 16, p.B), dim3(256), lds, st, p);
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    hipLaunchKernelGGL(conv_fewcout_f32_kernel<KS>, dim3((p.W + 15) / 16, (p.H + 15) / 16, p.B), dim3(256), 0, st, p);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int conv_pack_weights(sr_ctx* ctx, const float* hwio, const float* bias, int KS, int Cin, int Cout, int dtype,
                      ConvWeights* out, int rows_head) {
    if (dtype != SR_DTYPE_BF16 && dtype != SR_DTYPE_F32) return ctx->fail(SR_ERR_INVALID, "conv: dtype must be f32 or bf16");
    if (KS != 1 && KS != 3 && KS != 5 && KS != 9) return ctx->fail(SR_ERR_INVALID, "conv: kernel size must be 1,3,5 or 9");
    const int esz = dtype_size(dtype), E = 16 / esz;
    ConvWeights w;
    w.dtype = dtype; w.KS = KS; w.Cin = Cin; w.Cout = Cout;
    w.CoutP = round_up(Cout, 32);
    const int nb = w.CoutP / 32;
    w.NT = (nb % 2 == 0) ? 2 : (nb % 3 == 0 ? 3 : 1);
    if (KS == 5 && !(Cin <= E)) w.NT = 1;   // 25 taps of weights: keep the LDS stage small
    const bool as_rows = rows_head && dtype == SR_DTYPE_BF16 && KS == 3;
    if (Cin <= E && !as_rows) w.NT = 1;      // thin (RGB) inputs: one 32-cout block per workgroup (112 + 48 registers, 3 waves/SIMD);
                                             // re-reading the 3-channel input per cout block is cheap, 1 wave/SIMD at NT=3 was not (9x9: 1.8x)
    const int nct = nb / w.NT, ntap = KS * KS;
    w.thin = Cin <= E && !as_rows;
    // fp32, <= 4 couts, not thin: the VALU kernel (conv_fewcout_f32_kernel), weights [tap][CinP][4]
    w.few = (dtype == SR_DTYPE_F32 && Cout <= 4 && !w.thin && (KS == 3 || KS == 5)) ? 1 : 0;
    w.rows = (dtype == SR_DTYPE_BF16 && KS == 3 && !w.thin) ? 1 : 0;
    // bf16 1x1 with the whole weight matrix in 16 register fragments: the streaming kernel of conv_pw.hip (same fragment layout)
    {   // (register budget of the instantiations in conv_pw.hip: 4 waves/SIMD without spills)
        const int nb16 = round_up(Cout, 16) / 16, nch = round_up(Cin, 32) / 32;
        w.pw = (dtype == SR_DTYPE_BF16 && KS == 1 && !w.thin && nch <= 4 &&
                (nb16 <= 2 || (nb16 == 3 && nch <= 3) || (nb16 == 4 && nch == 1))) ? 1 : 0;
    }
    if (!w.thin && KS == 9) return ctx->fail(SR_ERR_INVALID, "conv: 9x9 supported for <= one 16-byte channel slice only");
    std::vector<char> host;
    auto put = [&](size_t idx, float v) {
        if (dtype == SR_DTYPE_F32) reinterpret_cast<float*>(host.data())[idx] = v;
        else reinterpret_cast<uint16_t*>(host.data())[idx] = f32_to_bf16_host(v);
    };
    auto W = [&](int tap, int ci, int co) -> float {
        if (tap >= ntap || ci >= Cin || co >= Cout) return 0.f;
        return hwio[((size_t)tap * Cin + ci) * Cout + co];
    };
    if (w.few) {
        w.CoutP = 4; w.NT = 1; w.KGPT = 0;
        w.CinP = round_up(Cin, 4);
        w.nchunks = w.CinP / 4;
        host.assign((size_t)ntap * w.CinP * 4 * sizeof(float), 0);
        for (int tap = 0; tap < ntap; ++tap)
            for (int ci = 0; ci < w.CinP; ++ci)
                for (int co = 0; co < 4; ++co) put(((size_t)tap * w.CinP + ci) * 4 + co, W(tap, ci, co));
    } else if (w.rows || w.pw) {
        w.CoutP = round_up(Cout, 16);
        const int nb16 = w.CoutP / 16;
        w.NT = w.pw ? nb16 : ((nb16 % 4 == 0) ? 4 : (nb16 % 2 == 0 ? 2 : 1));
        const int nct16 = nb16 / w.NT;
        w.KGPT = 1;
        w.CinP = round_up(Cin, 32);
        w.nchunks = w.CinP / 32;
        host.assign((size_t)nct16 * w.nchunks * ntap * w.NT * 1024, 0);
        size_t idx = 0;
        for (int ct = 0; ct < nct16; ++ct)
            for (int ch = 0; ch < w.nchunks; ++ch)
                for (int tap = 0; tap < ntap; ++tap)
                    for (int n = 0; n < w.NT; ++n)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j, ++idx)
                                put(idx, W(tap, ch * 32 + (lane >> 4) * 8 + j, (ct * w.NT + n) * 16 + (lane & 15)));
    } else if (w.thin) {
        w.CinP = E; w.KGPT = 0;
        const int KGT = (ntap + 1) / 2;
        w.nchunks = KGT;
        host.assign((size_t)nct * KGT * w.NT * 1024, 0);
        size_t idx = 0;
        for (int ct = 0; ct < nct; ++ct)
            for (int g = 0; g < KGT; ++g)
                for (int n = 0; n < w.NT; ++n)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < E; ++j, ++idx)
                            put(idx, W(2 * g + (lane >> 5), j, (ct * w.NT + n) * 32 + (lane & 31)));
    } else {
        w.KGPT = (KS == 1 && (round_up(Cin, 4 * E) * esz) % 128 == 0) ? 4 : 2;
        const int chunkE = w.KGPT * 2 * E;
        w.CinP = round_up(Cin, chunkE);
        w.nchunks = w.CinP / chunkE;
        host.assign((size_t)nct * w.nchunks * ntap * w.KGPT * w.NT * 1024, 0);
        size_t idx = 0;
        for (int ct = 0; ct < nct; ++ct)
            for (int ch = 0; ch < w.nchunks; ++ch)
                for (int tap = 0; tap < ntap; ++tap)
                    for (int kg = 0; kg < w.KGPT; ++kg)
                        for (int n = 0; n < w.NT; ++n)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < E; ++j, ++idx)
                                    put(idx, W(tap, ch * chunkE + kg * 2 * E + (lane >> 5) * E + j, (ct * w.NT + n) * 32 + (lane & 31)));
    }
    w.bytes = host.size();
    w.w = ctx->dalloc(w.bytes);
    if (!w.w) return SR_ERR_OOM;
    w.bias = static_cast<float*>(ctx->dalloc(sizeof(float) * w.CoutP));
    if (!w.bias) { ctx->dfree(w.w); return SR_ERR_OOM; }
    std::vector<float> hb(w.CoutP, 0.f);
    if (bias) for (int i = 0; i < Cout; ++i) hb[i] = bias[i];
    SR_HIP(ctx, hipMemcpy(w.w, host.data(), w.bytes, hipMemcpyHostToDevice));
    SR_HIP(ctx, hipMemcpy(w.bias, hb.data(), sizeof(float) * w.CoutP, hipMemcpyHostToDevice));
    *out = w;
    return SR_OK;
}

// ---- device-side packing (fp32): one thread per packed element decodes its (tap, cin, cout) exactly as the host loops above
struct PackPlan { int layout, ntap, Cin, Cout, CinP, NT, nchunks, KGPT, rot, CoutP; };

static __device__ __forceinline__ float pack_element_f32(const float* __restrict__ src, int64_t idx, const PackPlan& q) {
    int tap, ci, co;
    if (q.layout == 0) {                          // few: [tap][CinP][4]
        co = (int)(idx & 3);
        const int64_t t = idx >> 2;
        ci = (int)(t % q.CinP); tap = (int)(t / q.CinP);
    } else if (q.layout == 1) {                   // thin: [ct][g][n][lane][j], E = 4
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        int64_t t = idx >> 8;
        const int nn = (int)(t % q.NT); t /= q.NT;
        const int g = (int)(t % q.nchunks), ct = (int)(t / q.nchunks);          // nchunks = KGT
        tap = 2 * g + (lane >> 5); ci = j; co = (ct * q.NT + nn) * 32 + (lane & 31);
    } else {                                      // wide: [ct][ch][tap][kg][n][lane][j], E = 4
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        int64_t t = idx >> 8;
        const int nn = (int)(t % q.NT); t /= q.NT;
        const int kg = (int)(t % q.KGPT); t /= q.KGPT;
        tap = (int)(t % q.ntap); t /= q.ntap;
        const int ch = (int)(t % q.nchunks), ct = (int)(t / q.nchunks);
        ci = ch * (q.KGPT * 8) + kg * 8 + (lane >> 5) * 4 + j; co = (ct * q.NT + nn) * 32 + (lane & 31);
    }
    float v = 0.f;
    if (tap < q.ntap && ci < q.Cin && co < q.Cout)
        v = q.rot ? src[((int64_t)(q.ntap - 1 - tap) * q.Cout + co) * q.Cin + ci] : src[((int64_t)tap * q.Cin + ci) * q.Cout + co];
    return v;
}

__global__ void pack_weights_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, PackPlan q, const float* __restrict__ bias, float* __restrict__ bias_out) {
    // the zero-padded bias rides along (block 0): one launch per conv use less than a separate pad kernel
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < q.CoutP; i += blockDim.x) bias_out[i] = (bias && i < q.Cout) ? bias[i] : 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) dst[idx] = pack_element_f32(src, idx, q);
}

// Many convs' weights by ONE launch (sr_conv_prepack: a training step's ~800 per-use packs were 3.9 ms of 4.9-us launches): job j owns blocks [blk0, next job's blk0),
// PACK_JOB_ELEMS packed elements per block; a block finds its job by bisection of the table.
constexpr int PACK_JOB_ELEMS = 2048;
struct PackJob { const float* src; const float* bias; float* dst; float* bias_out; int64_t n; PackPlan q; int blk0; int pad_; };

__global__ void __launch_bounds__(256) pack_weights_f32_many_kernel(const PackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob& jb = jobs[lo];
    const PackPlan q = jb.q;
    const int lb = (int)blockIdx.x - jb.blk0;
    const float* __restrict__ src = jb.src;
    float* __restrict__ dst = jb.dst;
    const int64_t n = jb.n;
    if (lb == 0) {
        const float* bias = jb.bias;
        float* bo = jb.bias_out;
        for (int i = threadIdx.x; i < q.CoutP; i += 256) bo[i] = (bias && i < q.Cout) ? bias[i] : 0.f;
    }
    const int64_t base = (int64_t)lb * PACK_JOB_ELEMS;
#pragma unroll 4
    for (int i = threadIdx.x; i < PACK_JOB_ELEMS; i += 256)
        if (base + i < n) dst[base + i] = pack_element_f32(src, base + i, q);
}

// the metadata of conv_pack_weights for dtype f32 (no allocation): kernel family, padded sizes, packed element count
static int conv_plan_f32(sr_ctx* ctx, int KS, int Cin, int Cout, int rot, ConvWeights* out, PackPlan* plan, int64_t* count) {
    if (KS != 1 && KS != 3 && KS != 5 && KS != 9) return ctx->fail(SR_ERR_INVALID, "conv: kernel size must be 1,3,5 or 9");
    const int E = 4, ntap = KS * KS;
    ConvWeights w;
    w.dtype = SR_DTYPE_F32; w.KS = KS; w.Cin = Cin; w.Cout = Cout;
    w.CoutP = round_up(Cout, 32);
    const int nb = w.CoutP / 32;
    w.NT = (nb % 2 == 0) ? 2 : (nb % 3 == 0 ? 3 : 1);
    if (KS == 5 && !(Cin <= E)) w.NT = 1;
    if (Cin <= E) w.NT = 1;
    const int nct = nb / w.NT;
    w.thin = Cin <= E;
    w.few = (Cout <= 4 && !w.thin && (KS == 3 || KS == 5)) ? 1 : 0;
    w.rows = 0; w.pw = 0;
    if (!w.thin && KS == 9) return ctx->fail(SR_ERR_INVALID, "conv: 9x9 supported for <= one 16-byte channel slice only");
    int layout;
    int64_t n;
    if (w.few) {
        layout = 0;
        w.CoutP = 4; w.NT = 1; w.KGPT = 0;
        w.CinP = round_up(Cin, 4);
        w.nchunks = w.CinP / 4;
        n = (int64_t)ntap * w.CinP * 4;
    } else if (w.thin) {
        layout = 1;
        w.CinP = E; w.KGPT = 0;
        w.nchunks = (ntap + 1) / 2;
        n = (int64_t)nct * w.nchunks * w.NT * 256;
    } else {
        layout = 2;
        w.KGPT = (KS == 1 && (round_up(Cin, 4 * E) * 4) % 128 == 0) ? 4 : 2;
        const int chunkE = w.KGPT * 2 * E;
        w.CinP = round_up(Cin, chunkE);
        w.nchunks = w.CinP / chunkE;
        n = (int64_t)nct * w.nchunks * ntap * w.KGPT * w.NT * 256;
    }
    w.bytes = (size_t)n * 4;
    *out = w;
    *plan = PackPlan{layout, ntap, Cin, Cout, w.CinP, w.NT, w.nchunks, w.KGPT, rot, w.CoutP};
    *count = n;
    return SR_OK;
}

int conv_pack_weights_dev(sr_ctx* ctx, const float* d_hwio, const float* d_bias, int KS, int Cin, int Cout, int rot, ConvWeights* out, hipStream_t st) {
    ConvWeights w;
    PackPlan plan;
    int64_t n;
    const int rc = conv_plan_f32(ctx, KS, Cin, Cout, rot, &w, &plan, &n);
    if (rc) return rc;
    if (!ctx->pack_cache.empty()) {                                   // sr_conv_prepack: this use was packed with the step's other weights
        const auto it = ctx->pack_cache.find(sr_ctx::PackKey{d_hwio, d_bias, KS, Cin, Cout, rot});
        if (it != ctx->pack_cache.end()) {
            w.w = it->second.first; w.bias = it->second.second;
            *out = w;
            return SR_OK;
        }
    }
    w.w = ctx->arena(ctx->dev_w, w.bytes, st);
    w.bias = static_cast<float*>(ctx->arena(ctx->dev_b, sizeof(float) * round_up(w.CoutP, 64), st));
    if (!w.w || !w.bias) return SR_ERR_OOM;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_f32_kernel, dim3(blocks), dim3(256), 0, st, d_hwio, static_cast<float*>(w.w), n, plan, d_bias, w.bias);
    SR_HIP(ctx, hipGetLastError());
    *out = w;
    return SR_OK;
}

int conv_prepack_dev(sr_ctx* ctx, const sr_pack_desc* descs, int n, hipStream_t st) {
    ctx->pack_cache.clear();                                          // whatever happens below, no stale pack survives this call
    if (n <= 0) return SR_OK;
    std::vector<PackJob> jobs((size_t)n);
    std::vector<sr_ctx::PackKey> keys((size_t)n);
    size_t wbytes = 0, bbytes = 0;
    int64_t blocks = 0;
    std::vector<size_t> woff((size_t)n), boff((size_t)n);
    for (int i = 0; i < n; ++i) {
        const sr_pack_desc& d = descs[i];
        if (!d.w || d.Cin <= 0 || d.Cout <= 0) return ctx->fail(SR_ERR_INVALID, "conv prepack: null kernel or bad shape");
        ConvWeights w;
        PackJob& jb = jobs[(size_t)i];
        const int rc = conv_plan_f32(ctx, d.K, d.Cin, d.Cout, d.rot ? 1 : 0, &w, &jb.q, &jb.n);
        if (rc) return rc;
        jb.src = d.w; jb.bias = d.bias; jb.blk0 = (int)blocks; jb.pad_ = 0;
        blocks += (jb.n + PACK_JOB_ELEMS - 1) / PACK_JOB_ELEMS;
        woff[(size_t)i] = wbytes; boff[(size_t)i] = bbytes;
        wbytes += (w.bytes + 255) & ~(size_t)255;
        bbytes += (sizeof(float) * (size_t)round_up(w.CoutP, 64) + 255) & ~(size_t)255;
        keys[(size_t)i] = sr_ctx::PackKey{d.w, d.bias, d.K, d.Cin, d.Cout, d.rot ? 1 : 0};
    }
    if (blocks > 0x7fffffff) return ctx->fail(SR_ERR_INVALID, "conv prepack: too many elements for one launch");
    char* wb = static_cast<char*>(ctx->arena(ctx->pack_w, wbytes, st));
    char* bb = static_cast<char*>(ctx->arena(ctx->pack_b, bbytes, st));
    const size_t tbytes = sizeof(PackJob) * (size_t)n;
    void* tab = ctx->arena(ctx->pack_tab, tbytes, st);
    if (!wb || !bb || !tab) return SR_ERR_OOM;
    for (int i = 0; i < n; ++i) {
        jobs[(size_t)i].dst = reinterpret_cast<float*>(wb + woff[(size_t)i]);
        jobs[(size_t)i].bias_out = reinterpret_cast<float*>(bb + boff[(size_t)i]);
    }
    // the table is uploaded when it differs from the one on the device (a trainer's list is the same every step: pointers into its parameter bucket)
    if (ctx->pack_tab_host.size() != tbytes || ctx->pack_tab_dev != tab || memcmp(ctx->pack_tab_host.data(), jobs.data(), tbytes) != 0) {
        SR_HIP(ctx, hipStreamSynchronize(st));                         // an earlier launch may still be reading the old table
        SR_HIP(ctx, hipMemcpy(tab, jobs.data(), tbytes, hipMemcpyHostToDevice));
        ctx->pack_tab_host.assign(reinterpret_cast<const char*>(jobs.data()), reinterpret_cast<const char*>(jobs.data()) + tbytes);
        ctx->pack_tab_dev = tab;
    }
    hipLaunchKernelGGL(pack_weights_f32_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const PackJob*>(tab), n);
    SR_HIP(ctx, hipGetLastError());
    for (int i = 0; i < n; ++i) ctx->pack_cache[keys[(size_t)i]] = {jobs[(size_t)i].dst, jobs[(size_t)i].bias_out};
    return SR_OK;
}

void conv_free_weights(sr_ctx* ctx, ConvWeights* w) {
    if (w->w) ctx->dfree(w->w);
    if (w->bias) ctx->dfree(w->bias);
    w->w = nullptr; w->bias = nullptr;
}

int conv_launch(sr_ctx* ctx, const ConvWeights& w, TensorView x, int B, int H, int W, void* y, int64_t y_cs, int y_coff,
                const ConvEpilogue& ep, hipStream_t st) {
    return conv_launch(ctx, w, x, B, H, W, TensorView{y, y_cs, y_coff}, ep, st);
}

int conv_launch(sr_ctx* ctx, const ConvWeights& w, TensorView x, int B, int H, int W, TensorView yv, const ConvEpilogue& ep, hipStream_t st) {
    void* y = const_cast<void*>(yv.p);
    const int64_t y_cs = yv.cs;
    const int y_coff = yv.coff;
    if (B <= 0 || H <= 0 || W <= 0) return ctx->fail(SR_ERR_INVALID, "conv: empty tensor");
    const int esz = dtype_size(w.dtype);
    if ((x.cs * esz) % 16 != 0 || (x.coff * esz) % 16 != 0 || ((uintptr_t)x.p % 16) != 0)
        return ctx->fail(SR_ERR_INVALID, "conv: input view must be 16-byte aligned per pixel");
    if (x.cs - x.coff < w.CinP) return ctx->fail(SR_ERR_INVALID, "conv: input view exposes fewer channels than the packed Cin");
    if ((int64_t)H * W * x.cs >= (int64_t)1 << 31) return ctx->fail(SR_ERR_INVALID, "conv: image too large for 32-bit offsets");
    const int r = ep.d2s_r < 1 ? 1 : ep.d2s_r;
    if (w.Cout % (r * r) != 0) return ctx->fail(SR_ERR_INVALID, "conv: Cout not divisible by d2s block^2");
    // effective strides of a view (conv_common.h): NHWC or row-blocked
    auto strides = [&](const TensorView& v, int width, int64_t* cs, int* ps, int* rs) {
        if (v.blk) { *cs = 32; *ps = width * 32; *rs = (int)(width * v.cs); }
        else { *cs = v.cs; *ps = 32; *rs = (int)(width * v.cs); }
    };
    for (const TensorView* v : {(const TensorView*)&x, (const TensorView*)&yv, &ep.skip1, &ep.skip2})
        if (v->p && v->blk && (v->cs % 32 != 0 || v->coff % (v == &yv ? 4 : 32) != 0))      // the output may start inside a block (dense blocks of 8 / 16 / 24 growth channels): a lane stores 4 couts, choff() places them
            return ctx->fail(SR_ERR_INVALID, "conv: a row-blocked view needs 32-channel granularity (4 for the output's first channel)");
    const bool any_blk = x.blk || yv.blk || (ep.skip1.p && ep.skip1.blk) || (ep.skip2.p && ep.skip2.blk);
    if (w.pw && r > 1) return ctx->fail(SR_ERR_INVALID, "conv: 1x1 with depth_to_space is not built");
    if (any_blk && !w.rows) return ctx->fail(SR_ERR_INVALID, "conv: only the bf16 3x3 kernel handles row-blocked views");
    if (yv.blk && r > 1) return ctx->fail(SR_ERR_INVALID, "conv: depth_to_space writes NHWC only");
    ConvParams p;
    p.in = static_cast<const char*>(x.p); p.in_coff = x.coff; strides(x, W, &p.in_cs, &p.in_ps, &p.in_rs);
    p.w = static_cast<const char*>(w.w); p.bias = w.bias;
    p.out = static_cast<char*>(y); p.out_coff = y_coff; strides(yv, W * r, &p.out_cs, &p.out_ps, &p.out_rs);
    p.out_f32 = (ep.out_f32 || w.dtype == SR_DTYPE_F32) ? 1 : 0;
    p.s1 = static_cast<const char*>(ep.skip1.p); p.s1_coff = ep.skip1.coff; p.beta1 = ep.beta1; strides(ep.skip1, W, &p.s1_cs, &p.s1_ps, &p.s1_rs);
    p.s2 = static_cast<const char*>(ep.skip2.p); p.s2_coff = ep.skip2.coff; p.beta2 = ep.beta2; strides(ep.skip2, W, &p.s2_cs, &p.s2_ps, &p.s2_rs);
    if (!p.s1 && p.s2) {   // a lone skip is skip 1
        p.s1 = p.s2; p.s1_cs = p.s2_cs; p.s1_coff = p.s2_coff; p.s1_ps = p.s2_ps; p.s1_rs = p.s2_rs; p.beta1 = p.beta2; p.s2 = nullptr;
    }
    p.alpha = ep.alpha; p.act = ep.act; p.clip = ep.clip01; p.r = r; p.Cd = w.Cout / (r * r);
    p.B = B; p.H = H; p.W = W; p.Cout = w.Cout; p.nchunks = w.nchunks; p.tilesX = p.tilesY = 0;
    p.dbg = ctx->stamp_buf;
    p.f2w = nullptr; p.f2part = nullptr; p.f2c = 0;
    p.pjw = nullptr; p.pjbias = nullptr; p.pjout = nullptr; p.pj_cs = 0; p.pj_coff = 0; p.pj_rs = 0; p.pj_nblk = 0;
    p.cell_h = ep.cell_h; p.cell_w = ep.cell_w;
    p.splitk_ok = ep.allow_splitk;
    p.pw2w = nullptr; p.pw2bias = nullptr; p.pw2_cout = 0; p.pw2_act = 0;
    p.plout = nullptr; p.pl_cs = 0; p.pl_coff = 0; p.pl_gx = p.pl_ch = p.pl_cw = p.pl_Wv = 0;
    const int osz = p.out_f32 ? 4 : esz;
    bool vec = (y_cs % 4 == 0) && (y_coff % 4 == 0) && ((uintptr_t)y % (4 * osz) == 0) && (p.Cd % 4 == 0);
    if (p.s1) vec = vec && (p.s1_cs % 4 == 0) && (p.s1_coff % 4 == 0) && ((uintptr_t)p.s1 % (4 * esz) == 0);
    if (p.s2) vec = vec && (p.s2_cs % 4 == 0) && (p.s2_coff % 4 == 0) && ((uintptr_t)p.s2 % (4 * esz) == 0);
    p.vec = vec ? 1 : 0;
    if (any_blk && !(vec && w.Cout % 4 == 0 && ep.act != SR_ACT_TANH))   // conv_rows' generic per-element epilogue is NHWC-only
        return ctx->fail(SR_ERR_INVALID, "conv: row-blocked views need the vector epilogue (aligned views, Cout % 4 == 0, no tanh)");
    // a dense block adds its own input back: when a skip is exactly input channels [0, Cout) of this conv, conv_rows folds it in
    // from the LDS image of those channels while they are staged, instead of reading it again in the epilogue (conv_rows.hip)
    p.skip_lds = 0; p.skip_scale = 0.f;
    if (w.rows && w.NT == 4 && w.Cout % 16 == 0 && w.Cin >= w.Cout && ep.act == SR_ACT_LINEAR && ep.alpha != 0.f && r == 1 && vec) {
        if (p.s2 && p.s2 == p.in && p.s2_cs == p.in_cs && p.s2_ps == p.in_ps && p.s2_coff == p.in_coff) { p.skip_lds = 2; p.skip_scale = p.beta2 / p.alpha; }
        else if (p.s1 && p.s1 == p.in && p.s1_cs == p.in_cs && p.s1_ps == p.in_ps && p.s1_coff == p.in_coff) { p.skip_lds = 1; p.skip_scale = p.beta1 / p.alpha; }
    }
    if ((ep.cell_h || ep.cell_w) && !(w.rows && vec && w.Cout % 4 == 0 && ep.act != SR_ACT_TANH && r == 1 && !ep.pj && !ep.f2 && ep.cell_h >= 2 && ep.cell_w >= 2))
        return ctx->fail(SR_ERR_INVALID, "conv: separator masks need the bf16 3x3 kernel's vector epilogue");
    if (ep.pool_out.p) {
        if (!(w.rows && w.NT == 4 && w.CoutP == w.Cout && r == 1 && !p.s1 && !p.s2 && !ep.clip01 && ep.act != SR_ACT_TANH && !ep.pj && !ep.f2 && !ep.cell_h && !ep.cell_w && !ep.pool_out.blk &&
              ep.pool_out.cs % 8 == 0 && ep.pool_out.coff % 8 == 0 && ((uintptr_t)ep.pool_out.p % 16) == 0 && ep.pool_out.cs - ep.pool_out.coff >= w.Cout && H >= 2 && W >= 2))
            return ctx->fail(SR_ERR_INVALID, "conv: the fused max-pool follows a bf16 3x3 conv in 64-cout tiles without skips");
        p.plout = static_cast<char*>(const_cast<void*>(ep.pool_out.p)); p.pl_cs = ep.pool_out.cs; p.pl_coff = ep.pool_out.coff;
        p.pl_gx = ep.pool_grid.gx; p.pl_ch = ep.pool_grid.ch; p.pl_cw = ep.pool_grid.cw; p.pl_Wv = ep.pool_grid.Wv;
    }
    if (ep.pj) {
        // every output pixel's 64 channels come from one workgroup: a 64-cout conv, or depth_to_space of 64-channel sub-pixels
        if (!(w.rows && w.NT == 4 && p.Cd == 64 && w.CoutP == w.Cout && !yv.blk && !p.s2 && !ep.clip01 && ep.act != SR_ACT_TANH && vec && !ep.f2 && ep.pj->a &&
              ep.pj->nblk >= 1 && ep.pj->nblk <= 3 && ep.pj_out.p && !ep.pj_out.blk && ep.pj_out.cs - ep.pj_out.coff >= 16 * ep.pj->nblk && ep.pj_out.cs % 4 == 0 &&
              ep.pj_out.coff % 4 == 0 && (int64_t)H * r * W * r * ep.pj_out.cs < ((int64_t)1 << 31)))
            return ctx->fail(SR_ERR_INVALID, "conv: the fused 1x1 projection follows a bf16 3x3 conv whose output pixels have 64 channels (NHWC, at most one skip)");
        p.pjw = static_cast<const char*>(ep.pj->a); p.pjbias = ep.pj->bias; p.pj_nblk = ep.pj->nblk;
        p.skip_lds = 0; p.skip_scale = 0.f;                       // the projection epilogue adds skip 1 itself
        p.pjout = static_cast<char*>(const_cast<void*>(ep.pj_out.p)); p.pj_cs = ep.pj_out.cs; p.pj_coff = ep.pj_out.coff; p.pj_rs = (int)(W * r * ep.pj_out.cs);
    }
    if (ep.pw2) {
        if (!(w.dtype == SR_DTYPE_F32 && w.thin && !w.few && w.NT == 1 && w.CoutP == 96 && ep.pw2->a && ep.pw2->cin == w.Cout && ep.pw2->cout >= 1 && ep.pw2->cout <= 32 && r == 1 &&
              !p.s1 && !p.s2 && !ep.clip01 && !ep.pj && !ep.f2 && !ep.pool_out.p && !yv.blk && yv.cs - yv.coff >= ep.pw2->cout))
            return ctx->fail(SR_ERR_INVALID, "conv: the fused 1x1 follows an fp32 thin conv that owns all its couts in one workgroup, without skips");
        p.pw2w = ep.pw2->a; p.pw2bias = ep.pw2->bias; p.pw2_cout = ep.pw2->cout; p.pw2_act = ep.pw2->act;
        p.Cd = ep.pw2->cout;                                       // what is stored has pw2_cout channels: the vector-store test below is about them
    }
    if (ep.f2) {
        if (!(w.rows && w.NT == 4 && w.Cout == 64 && w.CoutP == 64 && r == 1 && !p.s1 && !p.s2 && !ep.clip01 && ep.act != SR_ACT_TANH && ep.f2->a && ep.f2_part))
            return ctx->fail(SR_ERR_INVALID, "conv: the fused RGB tail follows a bf16 3x3 conv to 64 channels without skips");
        p.f2w = static_cast<const char*>(ep.f2->a); p.f2part = ep.f2_part; p.f2c = ep.f2->c2;
    }
    const int nct = w.CoutP / 32 / w.NT;
    const bool stream = (ctx->chain_mask & 128) != 0 && conv_stream_supported(w, p);
    int rec = -1;
    if (ctx->prof && ep.f2) {
        const double px = (double)B * H * W;
        rec = ctx->prof_open("conv_rows_rgbtail<bf16,64->64->rgb>", 2.0 * px * 9 * w.Cin * (w.Cout + ep.f2->c2),
                             px * w.Cin * esz + (double)rgbtail_partial_bytes(B, H, W), st);
    } else if (ctx->prof && ep.pool_out.p) {
        const double px = (double)B * H * W;
        rec = ctx->prof_open("conv_rows_pool<bf16,k3,nt4+maxpool2>", 2.0 * px * 9.0 * w.Cin * w.Cout, px * ((double)w.Cin * esz + 0.25 * w.Cout * esz), st);
    } else if (ctx->prof && ep.pj) {
        const double px = (double)B * H * W;
        rec = ctx->prof_open("conv_rows_proj<bf16,k3,nt4+1x1>", 2.0 * px * (9.0 * w.Cin * w.Cout + (double)w.Cout * 16 * ep.pj->nblk),
                             px * ((double)w.Cin * esz + (double)w.Cout * osz + (p.s1 ? (double)w.Cout * esz : 0.0) + (double)r * r * 16 * ep.pj->nblk * esz), st);
    } else if (ctx->prof && ep.pw2) {
        const double px = (double)B * H * W;
        rec = ctx->prof_open("conv_thin_pw2<f32,k9,3->96->32>", 2.0 * px * ((double)w.KS * w.KS * w.Cin * w.Cout + (double)w.Cout * ep.pw2->cout),
                             px * ((double)w.CinP * esz + (double)ep.pw2->cout * osz), st);
    } else if (ctx->prof) {
        char nm[96];
        snprintf(nm, sizeof nm, "conv_%s<%s,k%d,kg%d,nt%d>", w.few ? "few" : stream ? "stream" : w.rows ? "rows" : (w.pw ? "pw" : (w.thin ? "thin" : "wide")),
                 w.dtype == SR_DTYPE_BF16 ? "bf16" : "f32", w.KS, w.KGPT, w.NT);
        const double px = (double)B * H * W;
        double bytes = px * ((double)w.Cin * esz + (double)w.Cout * osz);
        if (p.s1) bytes += px * w.Cout * esz;
        if (p.s2) bytes += px * w.Cout * esz;
        rec = ctx->prof_open(nm, 2.0 * px * w.KS * w.KS * w.Cin * w.Cout, bytes, st);
    }
    const int rc = w.few ? (w.KS == 3 ? launch_fewcout<3>(ctx, p, st) : launch_fewcout<5>(ctx, p, st))
                   : stream ? conv_stream_launch(ctx, w, p, st) : w.rows ? conv_rows_launch(ctx, w, p, st) : w.pw ? conv_pw_launch(ctx, w, p, st)
                          : ((w.dtype == SR_DTYPE_BF16) ? dispatch<bf16_t>(ctx, w, p, nct, st) : dispatch<float>(ctx, w, p, nct, st));
    ctx->prof_close(rec, st);
    return rc;
}

// A operands of the fused 1x1 behind the fp32 thin kernel (conv_thin_kernel, PW2): step (n, i) of an M-block's MFMA chain multiplies, in lane l,
// W2[cin = 32 n + 8 (i / 4) + 4 (l / 32) + (i % 4)][cout = l % 32] with the activation the same lane holds in element i of accumulator block n
int pw2_pack_weights(sr_ctx* ctx, const float* w, const float* bias, int cin, int cout, int act, Pw2Weights* out) {
    if (cin < 32 || cin % 32 != 0 || cin > 96 || cout < 1 || cout > 32) return ctx->fail(SR_ERR_INVALID, "fused 1x1: 32, 64 or 96 input and 1..32 output channels");
    const int NT = cin / 32;
    std::vector<float> host((size_t)NT * 16 * 64, 0.f);
    for (int n = 0; n < NT; ++n)
        for (int i = 0; i < 16; ++i)
            for (int l = 0; l < 64; ++l) {
                const int c = 32 * n + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), co = l & 31;
                if (co < cout) host[((size_t)n * 16 + i) * 64 + l] = w[(size_t)c * cout + co];
            }
    Pw2Weights pw;
    pw.cin = cin; pw.cout = cout; pw.act = act;
    pw.a = static_cast<float*>(ctx->dalloc(host.size() * sizeof(float)));
    if (!pw.a) return SR_ERR_OOM;
    pw.bias = static_cast<float*>(ctx->dalloc(32 * sizeof(float)));
    if (!pw.bias) { ctx->dfree(pw.a); return SR_ERR_OOM; }
    float hb[32] = {0.f};
    if (bias) for (int i = 0; i < cout; ++i) hb[i] = bias[i];
    SR_HIP(ctx, hipMemcpy(pw.a, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    SR_HIP(ctx, hipMemcpy(pw.bias, hb, sizeof hb, hipMemcpyHostToDevice));
    *out = pw;
    return SR_OK;
}

void pw2_free_weights(sr_ctx* ctx, Pw2Weights* w) {
    if (w->a) ctx->dfree(w->a);
    if (w->bias) ctx->dfree(w->bias);
    *w = Pw2Weights{};
}
