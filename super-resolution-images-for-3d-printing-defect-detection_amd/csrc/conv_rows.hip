// conv_rows.hip -- bf16 3x3 SAME convolution, "row-sliding" implicit GEMM on v_mfma_f32_16x16x32_bf16.
//
// This is the kernel the ESRGAN/EDSR/VGG trunks spend their time in.  The first-generation kernel
// (conv.hip) reads one pixel fragment from LDS per MFMA when Cout = 32, which makes the LDS -- not the
// matrix cores -- the bound.  Here every pixel fragment is used by all three ky taps:
//
//   wave  = a strip of R output rows x 16 columns; workgroup = 4 waves stacked vertically (4R x 16);
//   MFMA  = 16 couts x 16 pixels x 32 channels; A = weight fragment, B = pixel fragment, so a lane
//           owns one pixel column and 4 consecutive couts (8-byte NHWC stores);
//   loop  = for each 32-channel chunk, for kx in 0..2: hold the 3 (ky) x NB16 weight fragments in
//           registers, walk the R+2 input rows of the strip once; the fragment of input row yi feeds
//           output rows yi, yi-1, yi-2 (ky = 0, 1, 2).
//   LDS reads per 16-cycle MFMA: (R+2)/(3*R*NB16) pixel + 1/R weight fragments (0.5 at R=4, Cout=32;
//   the old kernel: 1.33 per 32-cycle MFMA).
//
// Tile sizes are set by occupancy, not by MFMA efficiency: these layers stream their input from HBM once and run
// against the memory system, so what pays is workgroups in flight per CU.  16x16 tiles (R=4) at 32 or 16
// couts per workgroup: 39 KB of LDS, <= 128 VGPRs -> 4 workgroups/CU; 12x16 tiles (R=3) at 64 couts: 52 KB, <= 168 VGPRs
// -> 3 workgroups/CU.  (24x16 tiles at 2-3 workgroups/CU had fewer LDS reads per MFMA and were 5-7 % slower.)
//
// LDS image of the input halo tile: pixel-major, 64 B per pixel per chunk, no padding; 16-byte slice q
// of pixel index t (t = row*18 + col) sits at t*64 + ((q*16) ^ ((t & 4) << 3)).  With that XOR the 16
// lanes a ds_read_b128 services together hit 16 distinct 16-byte bank slots for every tap offset
// (checked exhaustively over all shifts; SQ_LDS_BANK_CONFLICT = 0 measured), and a staging wave still
// writes 1 KiB contiguously.  Weights are host-packed lane-linear per (chunk, tap, cout-block-of-16): a
// straight copy, conflict free.
//
// Prologue and epilogue are kept to a few hundred issue slots per wave (32-bit offsets from wave-uniform
// 64-bit bases, branch-free activation, biases prefetched, skips loaded in bulk): in-kernel stamps showed
// they, not the MFMAs, held the SIMDs in the first version.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_rows_epi.h"

namespace {

using namespace convk;

__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

#define STAMP_AT(i) do { if (STAMP && tid == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// Fused RGB tail (FUSE2; 64 couts per workgroup only).  The generator ends in final_conv1 (64 -> 64, ReLU) -> final_conv2 (64 -> 3, tanh)
// (ESRGAN_model.py:339-341): layer by layer the 64-channel image at the output resolution is written and read back once -- 66 GB per
// bench step, as much as final_conv1's own input -- for a conv that is 4 % of final_conv1's FLOP.  A 3x3 conv to C2 <= 3 channels is a 1x1
// conv to 9 C2 <= 27 "tap channels" followed by a shifted sum,
//     out[y][x][co] = sum_taps z[y + ky - 1][x + kx - 1][tap][co],   z[p][tap][co] = sum_c W2[tap][c][co] h[p][c],
// and z of a pixel needs that pixel's h only -- which the epilogue holds in registers, already in the MFMA's B-operand shape: lane (px, q)
// has channels 16 n + 4 q + e of pixel px for cout blocks n = 0..3, i.e. for a "channel half" (n = 2 half, 2 half + 1) eight bf16 values; the
// K ordering of an MFMA is free as long as the A fragment (host-packed, rgbtail_pack_weights) uses the same one.  So: 4 MFMAs per output row
// and wave give z (fp32, 32 tap channels); z goes to LDS; thread i < 14 * 18 sums, for position i of the tile's halo'd region, the taps whose
// source pixel lies inside the tile, and stores 3 floats.  h is never written.  rgbtail_finish_kernel adds the <= 4 tiles' partial sums of
// an output pixel in a fixed order, then bias, tanh, store.  h enters the 1x1 as a bf16 hi + lo pair (round 4; the layer-by-layer path stores and
// re-reads it as ONE bf16 value), the sums are fp32 throughout.
template <int R>
__device__ __forceinline__ void rows_fuse2(const ConvParams& p, f32x4 (&acc)[R][4], const f32x4 (&biasv)[4], char* smem, int64_t tile, int y0, int x0,
                                           int wave, int lane, int px, int q, int tid) {
    constexpr int TH = 4 * R, RW = 18, ZS = 36;             // ZS: floats per pixel in LDS (32 tap channels + 4: 16-lane ds_write_b128 groups hit 64 distinct banks)
    const float slope = p.act == SR_ACT_RELU ? 0.f : (p.act == SR_ACT_LRELU ? 0.2f : 1.f);
    const float alpha = p.alpha;
    bf16x8 wa[2][2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wa[tb][hf] = *reinterpret_cast<const bf16x8*>(p.f2w + ((tb * 2 + hf) * 64 + lane) * 16);
    const int ox = x0 + px;
    f32x4 z[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool live = y0 + wave * R + r < p.H && ox < p.W;             // pixels of a ragged tile beyond the image contribute nothing
        // h enters the 1x1 product as hi + lo, two bf16 values (lo = the part of the fp32 value the first rounding dropped): the activation is
        // never stored, so nothing asks for its rounding to 8 mantissa bits, and of the output end's storage roundings this one weighed most in
        // the image (round 4: 1.6 dB of the bf16 path's noise floor against the fp32 graph; 4 more MFMAs per row beside final_conv1's 72)
        bf16x8 hb[2], lb[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[r][2 * hf + u][e] + biasv[2 * hf + u][e];
                    v = fmaxf(v, v * slope) * alpha;
                    const bf16_t hi = (bf16_t)v;
                    hb[hf][4 * u + e] = live ? hi : (bf16_t)0.f;
                    lb[hf][4 * u + e] = live ? (bf16_t)(v - (float)hi) : (bf16_t)0.f;
                }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            z[r][tb] = mma16(wa[tb][0], lb[0], f32x4{0.f, 0.f, 0.f, 0.f});
            z[r][tb] = mma16(wa[tb][1], lb[1], z[r][tb]);
            z[r][tb] = mma16(wa[tb][0], hb[0], z[r][tb]);
            z[r][tb] = mma16(wa[tb][1], hb[1], z[r][tb]);
        }
    }
    __syncthreads();                                                       // every wave is done with the input image and the weights
    float* zl = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) *reinterpret_cast<f32x4*>(zl + ((wave * R + r) * 16 + px) * ZS + 16 * tb + 4 * q) = z[r][tb];
    __syncthreads();
    if (tid < (TH + 2) * RW) {
        const int ry = tid / RW, rx = tid - ry * RW, c2 = p.f2c;
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = ry + ky - 2;                                    // source pixel (tile-local) of tap (ky, kx) for output (y0 - 1 + ry, x0 - 1 + rx)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = rx + kx - 2;
                if ((unsigned)sy < (unsigned)TH && (unsigned)sx < 16u) {
                    const float* zp = zl + (sy * 16 + sx) * ZS + (ky * 3 + kx) * c2;
#pragma unroll
                    for (int co = 0; co < 3; ++co)
                        if (co < c2) s[co] += zp[co];
                }
            }
        }
        float* dst = p.f2part + (tile * ((TH + 2) * RW) + tid) * 3;
        dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2];
    }
}

// Fused 1x1 projection (64 couts per workgroup only).  SelfAttention (ESRGAN_model.py:48-56) opens with three 1x1 convs f / g / h of its
// 64-channel input, 48 output channels together; as a kernel of its own that is a pure stream (read 128 B, write 96 B per pixel: 7 ms per
// bench step).  The producing conv's epilogue has the pixel's 64 channels in registers in the MFMA's B-operand shape (rows_fuse2 above), so
// the projection is 2 MFMAs per 16-channel block and output row, taken from the bf16-rounded values that are being stored -- the attention
// input is read back by nobody but the attention's residual add.  With depth_to_space the cout tile of 64 is one sub-pixel's 64 channels,
// so the same holds at the up-sampled resolution.  Main stores as in rows_epilogue_fast<PAIR>; NHWC views only (host-checked).
template <int R, bool HAS1>
__device__ __forceinline__ void rows_epilogue_proj(const ConvParams& p, f32x4 (&acc)[R][4], const f32x4 (&biasv)[4], int b, int y0, int x0, int ct,
                                                   int wave, int lane, int px, int q) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int H = p.H, W = p.W, ox = x0 + px, oyw = y0 + wave * R;
    const int rows = min(R, H - oyw);
    if (rows <= 0) return;
    const bool col_ok = ox < W;
    const int oxc = col_ok ? ox : 0;
    const float slope = p.act == SR_ACT_RELU ? 0.f : (p.act == SR_ACT_LRELU ? 0.2f : 1.f);
    const float alpha = p.alpha, beta1 = p.beta1;
    const int rr = p.r, nblk = p.pj_nblk;
    const int si = rr > 1 ? ct / rr : 0, sj = rr > 1 ? ct - si * rr : 0;      // depth_to_space (DCR, Cd = 64): cout tile ct is sub-pixel (si, sj)
    // main output: lanes q and q ^ 1 trade halves, an even-q lane stores 8 couts of block 2s, an odd-q lane 8 of block 2s + 1
    const int lane_c = (q & 1) * 16 + 4 * (q & ~1);
    const int64_t orow0 = ((int64_t)b * H + oyw) * rr + si;                   // output row of tile row 0; rows advance by rr
    const int ooff = (ox * rr + sj) * (int)p.out_cs + p.out_coff + (rr > 1 ? 0 : ct * 64) + lane_c;
    const int poff = (ox * rr + sj) * (int)p.pj_cs + p.pj_coff + 4 * q;
    bf16x4 k1[HAS1 ? R : 1][4];
    if (HAS1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = (int64_t)b * H + (oyw + (r < rows ? r : 0));
#pragma unroll
            for (int n = 0; n < 4; ++n)
                k1[r][n] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(p.s1) + row * p.s1_rs + (oxc * (int)p.s1_cs + choff(p.s1_coff + ct * 64 + n * 16 + 4 * q, p.s1_ps)));
        }
    }
    bf16x8 wa[3][2];
    f32x4 pjb[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int nbc = nb < nblk ? nb : 0;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) wa[nb][hf] = *reinterpret_cast<const bf16x8*>(p.pjw + ((nbc * 2 + hf) * 64 + lane) * 16);
        pjb[nb] = *reinterpret_cast<const f32x4*>(p.pjbias + nbc * 16 + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r >= rows) break;                                                  // wave-uniform
        bf16x4 o[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 v = acc[r][n] + biasv[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope) * alpha;
            if (HAS1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += beta1 * (float)k1[r][n][e];
            }
            o[n] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
        bf16_t* const orow = reinterpret_cast<bf16_t*>(p.out) + (orow0 + (int64_t)r * rr) * p.out_rs;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x2 au = __builtin_bit_cast(u32x2, o[2 * s]), cu = __builtin_bit_cast(u32x2, o[2 * s + 1]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], cu[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], cu[1], false, false);
            const u32x4 ov = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
            if (col_ok) *reinterpret_cast<u32x4*>(orow + ooff + 32 * s) = ov;
        }
        // the projection of what was just stored: B fragment of channel half hf = the lane's 8 channels {16 (2 hf) + 4 q + e, 16 (2 hf + 1) + 4 q + e}
        bf16x8 hb[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 4; ++e) { hb[hf][e] = o[2 * hf][e]; hb[hf][4 + e] = o[2 * hf + 1][e]; }
        bf16_t* const prow = reinterpret_cast<bf16_t*>(p.pjout) + (orow0 + (int64_t)r * rr) * p.pj_rs;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            if (nb >= nblk) break;                                             // uniform
            f32x4 z = mma16(wa[nb][0], hb[0], pjb[nb]);
            z = mma16(wa[nb][1], hb[1], z);
            const bf16x4 zo = {(bf16_t)z[0], (bf16_t)z[1], (bf16_t)z[2], (bf16_t)z[3]};
            if (col_ok) *reinterpret_cast<bf16x4*>(prow + poff + 16 * nb) = zo;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Fused 2x2 max-pool (FUSE2 == 2; 64 couts per workgroup).  Every VGG16 block ends conv -> MaxPooling2D (VGG16_model.py:69-73): as two kernels the conv's
// full-resolution output is written and read back for a reduction that keeps a quarter of it.  The tile's rows pair up across waves (three rows per
// wave), so the bf16 outputs go through the tile's LDS (12 x 16 pixels x 128 B = 24 KiB): thread v < 384 takes the maximum of the four pixels of pooled
// position v / 8 for channel octet v % 8 and stores 16 bytes.  max of bf16 values is exact: the pooled tensor is the two-kernel path's bit for bit.
template <int R>
__device__ __forceinline__ void rows_pool2(const ConvParams& p, f32x4 (&acc)[R][4], const f32x4 (&biasv)[4], char* smem, int b, int y0, int x0, int ct, int wave,
                                           int px, int q, int tid) {
    constexpr int TH = 4 * R, PB = 128 + 16;                       // bytes per pixel in LDS (64 channels + 16: the 8-byte writes of a lane group spread over the banks)
    const float slope = p.act == SR_ACT_RELU ? 0.f : (p.act == SR_ACT_LRELU ? 0.2f : 1.f);
    const float alpha = p.alpha;
    __syncthreads();                                               // every wave is done with the input image and the weights
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 v = acc[r][n] + biasv[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope) * alpha;
            *reinterpret_cast<bf16x4*>(smem + ((wave * R + r) * 16 + px) * PB + (16 * n + 4 * q) * 2) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    __syncthreads();
    const int Ho = p.H >> 1, Wo = p.W >> 1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int v = tid + 256 * it;
        if (v >= (TH / 2) * 8 * 8) break;
        const int cv = v & 7, ppx = (v >> 3) & 7, pr = v >> 6;
        const int oy = (y0 >> 1) + pr, ox = (x0 >> 1) + ppx;
        if (oy >= Ho || ox >= Wo) continue;
        const char* s = smem + ((2 * pr) * 16 + 2 * ppx) * PB + cv * 16;
        const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(s), v01 = *reinterpret_cast<const bf16x8*>(s + PB);
        const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(s + 16 * PB), v11 = *reinterpret_cast<const bf16x8*>(s + 17 * PB);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(fmaxf((float)v00[e], (float)v01[e]), fmaxf((float)v10[e], (float)v11[e]));
        const int64_t pix = p.pl_gx == 0 ? ((int64_t)b * Ho + oy) * Wo + ox : ((int64_t)(b / p.pl_gx) * p.pl_ch + oy) * p.pl_Wv + (b % p.pl_gx) * p.pl_cw + ox;
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.plout) + pix * p.pl_cs + p.pl_coff + ct * 64 + cv * 8) = o;
    }
}

template <int NB16, int R, bool STAMP, int FUSE2 = 0>
__global__ void __launch_bounds__(256, NB16 == 4 ? (R > 3 ? 2 : 3) : 4) conv3_rows_kernel(ConvParams p) {
    constexpr int TH = 4 * R, TW = 16, PH = TH + 2, PW = TW + 2;
    constexpr int NPIX = PH * PW;
    constexpr int NIN = NPIX * 4;                 // 16-byte units per chunk
    constexpr int NINT = (NIN + 255) / 256;
    constexpr int LIN_BYTES = NPIX * 64;
    constexpr int WUNITS = 9 * NB16 * 64;         // 16-byte units of weights per chunk
    constexpr int NWT = (WUNITS + 255) / 256;     // weight units per thread
    constexpr bool WPRE = NB16 <= 1;              // prefetch next chunk's weights into registers when they fit the occupancy target's VGPR budget
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lin = smem;
    char* lw = smem + LIN_BYTES;
    float* lbias = reinterpret_cast<float*>(smem + LIN_BYTES + WUNITS * 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, q = lane >> 4;
    STAMP_AT(0);
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give every XCD a contiguous run
    // of the tile list -- the tiles of one image then share an L2 and their halo re-reads stay on chip.
    int t;
    {
        const int total = gridDim.x, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        const int qn = total >> 3, rn = total & 7;
        t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + k;
    }
    // the cout tiles of one spatial tile are neighbours in the order (same XCD, same time): the input they all stage
    // is fetched from HBM once and served from L2 to the others
    const int nct = p.tilesX >> 16;                          // packed by the launcher: tilesX = (nct << 16) | tiles in x
    const int tilesX = p.tilesX & 0xffff;
    const int ct = t % nct; t /= nct;
    const int tx = t % tilesX; t /= tilesX;
    const int ty = t % p.tilesY;
    const int b = t / p.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = p.H, W = p.W, in_cs = (int)p.in_cs, in_rs = p.in_rs, in_ps = p.in_ps;
    const bf16_t* inb = reinterpret_cast<const bf16_t*>(p.in) + (int64_t)b * H * in_rs + choff(p.in_coff, in_ps);   // wave-uniform

    int soff[NINT], doff[NINT];   // 32-bit element offset inside the image (-1: zero fill) / LDS byte offset (-1: none)
#pragma unroll
    for (int i = 0; i < NINT; ++i) {
        const int u = tid + 256 * i;
        const int pix = u >> 2, sl = u & 3;
        const int py = pix / PW, pxx = pix - py * PW;
        const int gy = y0 + py - 1, gx = x0 + pxx - 1;
        const bool live = u < NIN;
        const bool inside = live && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        doff[i] = live ? pix * 64 + ((sl * 16) ^ ((pix & 4) << 3)) : -1;
        soff[i] = inside ? gy * in_rs + gx * in_cs + sl * 8 : -1;
    }
    // Skip from LDS (p.skip_lds = 1 or 2): that skip tensor IS input channels [0, Cout) of this conv (a dense block adding its
    // own input back; linear activation, host-checked).  Cout block g then needs x channels 16g..16g+15 at the centre pixel,
    // which sit in the LDS image while chunk g/2 is staged: they are folded into the accumulators right there,
    //   alpha*(acc + b) + beta*x == alpha*((acc + (beta/alpha)*x) + b)   (up to fp32 rounding),
    // and the epilogue no longer reads that skip from HBM.
    const int nch = p.nchunks;
    constexpr bool SKIP_LDS_OK = NB16 == 4;      // only the 64-couts-per-workgroup variant carries the code (register budget)
    const bool skip_lds = SKIP_LDS_OK && p.skip_lds != 0;
    bf16x8 pre[NINT];
    f32x4 wpre[NWT];
    const char* wbase = p.w + (int64_t)ct * p.nchunks * (int64_t)(WUNITS * 16);
    // The prefetch loads are inline asm on purpose: hipcc otherwise waits for them (s_waitcnt vmcnt) right after issue --
    // it touches their destination registers early -- and the next chunk's HBM latency is no longer hidden under this
    // chunk's MFMAs (seen in the .s of the plain-C++ version: vmcnt(3..0) between the loads and the first MFMA).  An asm
    // load is invisible to the compiler's wait bookkeeping; land_all() below is the one explicit wait, placed right before
    // the registers are written to LDS, and it names every destination so nothing is scheduled across it.
    auto issue = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < NINT; ++i) {
            const int so = soff[i];
            const bf16_t* ptr = inb + (so >= 0 ? so + chunk * in_ps : 0);   // always a valid address; zero-select happens after landing
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre[i]) : "v"(ptr) : "memory");
        }
    };
    auto issue_w = [&](int chunk) {
        const char* wsrc = wbase + (int64_t)chunk * (WUNITS * 16);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int u = tid + 256 * i;
            const char* ptr = wsrc + (u < WUNITS ? u : 0) * 16;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wpre[i]) : "v"(ptr) : "memory");
        }
    };
    // Without register prefetch the chunk's weights go global -> LDS by LDS-DMA (global_load_lds_dwordx4; the packed layout is the
    // LDS layout, 1 KiB per wave-instruction): no VGPR round trip and none of the ds_write_b128 traffic (36 KB per chunk at 64
    // couts) through the VGPR -> LDS path.  Issued after the barrier that frees the weight buffer, landed by land_all()'s wait.
    constexpr bool WDMA = !WPRE;
    // Every wave issues exactly DMA_PER_WAVE instructions -- the counted vmcnt wait below relies on that number, and
    // tools/check_prefetch_hazards.py proves it on the control-flow graph: no branch around an issue.  Where the piece count is not a
    // multiple of 4 (18 pieces at 32 couts) the waves that run out re-fetch the last piece: same bytes to the same LDS address.
    constexpr int PIECES = WUNITS / 64, DMA_PER_WAVE = (PIECES + 3) / 4;
    auto dma_w = [&](int chunk) {
        const char* wsrc = wbase + (int64_t)chunk * (WUNITS * 16);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int piece = min(wv + 4 * i, PIECES - 1);   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + ((size_t)piece * 64 + lane) * 16),
                                             (__attribute__((address_space(3))) void*)(lw + piece * 1024), 16, 0, 0);
        }
    };
    auto land_all = [&]() {   // wait for every asm load in flight; "+v" on each destination orders all their uses behind the wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NINT; ++i) asm volatile("" : "+v"(pre[i]));
        if constexpr (WPRE) {
#pragma unroll
            for (int i = 0; i < NWT; ++i) asm volatile("" : "+v"(wpre[i]));
        }
    };
    auto write_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int u = tid + 256 * i;
            if (u < WUNITS) *reinterpret_cast<f32x4*>(lw + u * 16) = wpre[i];
        }
    };

    issue(0);
    if (WPRE) issue_w(0);
    // the workgroup's biases wait in LDS for the epilogue: no global latency there, and no registers held across the loop
    if (tid < NB16 * 16) lbias[tid] = p.bias[ct * NB16 * 16 + tid];

    f32x4 acc[R][NB16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < NB16; ++n) acc[r][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS byte address of the pixel fragment for (input row yi, kx): tt = tbase + yi*PW + kx
    const int tbase = wave * R * PW + px;
    auto xaddr = [&](int yi, int kx) {
        const int tt = tbase + yi * PW + kx;
        return tt * 64 + ((q * 16) ^ ((tt & 4) << 3));
    };
    STAMP_AT(1);

    for (int chunk = 0; chunk < nch; ++chunk) {
        __syncthreads();
        if (chunk < 6) STAMP_AT(2 + 2 * chunk);
        if constexpr (WDMA) {
            // the weight DMA just issued is younger than the input prefetch (issued a whole compute phase ago): wait for all but
            // this wave's DMA instructions (vmcnt counts in order), write the input image while the weights are still in flight
            dma_w(chunk);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
#pragma unroll
            for (int i = 0; i < NINT; ++i) asm volatile("" : "+v"(pre[i]));
        } else {
            land_all();
        }
#pragma unroll
        for (int i = 0; i < NINT; ++i)
            if (doff[i] >= 0) {
                bf16x8 z = {};
                *reinterpret_cast<bf16x8*>(lin + doff[i]) = soff[i] >= 0 ? pre[i] : z;
            }
        if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else write_w();
        __syncthreads();
        if (chunk < 6) STAMP_AT(3 + 2 * chunk);
        if (chunk + 1 < nch) {
            issue(chunk + 1);
            if (WPRE) issue_w(chunk + 1);
        }
        // software pipeline over (kx, yi): the fragment of the next input row is requested before this row's MFMAs
        bf16x8 xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(0, 0));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 wf[3][NB16];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int n = 0; n < NB16; ++n)
                    wf[ky][n] = *reinterpret_cast<const bf16x8*>(lw + ((ky * 3 + kx) * NB16 + n) * 1024 + lane * 16);
#pragma unroll
            for (int yi = 0; yi < R + 2; ++yi) {
                const bf16x8 xf = xcur;
                if (yi + 1 < R + 2) xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(yi + 1, kx));
                else if (kx + 1 < 3) xcur = *reinterpret_cast<const bf16x8*>(lin + xaddr(0, kx + 1));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yo = yi - ky;
                    if (yo >= 0 && yo < R) {
#pragma unroll
                        for (int n = 0; n < NB16; ++n) acc[yo][n] = mma16(wf[ky][n], xf, acc[yo][n]);
                    }
                }
            }
        }
        if constexpr (SKIP_LDS_OK) if (skip_lds) {
            const float sc = p.skip_scale;
#pragma unroll
            for (int n = 0; n < NB16; ++n) {
                const int g = ct * NB16 + n;                                     // wave-uniform
                if ((g >> 1) != chunk) continue;
                const int slice = (g & 1) * 2 + (q >> 1);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int tt = (wave * R + r + 1) * PW + px + 1;             // centre pixel of output (row r, column px)
                    const bf16x4 xk = *reinterpret_cast<const bf16x4*>(lin + tt * 64 + ((slice * 16) ^ ((tt & 4) << 3)) + (q & 1) * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][n][e] += sc * (float)xk[e];
                }
            }
        }
    }
    // nothing may be in flight past this point (there is not, on any path the host can produce: the last chunk issues no prefetch
    // and nchunks >= 1) -- stated as an instruction so that the hazard checker can prove it path-insensitively; free when idle
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_AT(14);

    // ---- epilogue (conv_rows_epi.h)
    f32x4 biasv[NB16];
#pragma unroll
    for (int n = 0; n < NB16; ++n) biasv[n] = *reinterpret_cast<const f32x4*>(lbias + n * 16 + 4 * q);
    if constexpr (FUSE2 == 2) {
        static_assert(FUSE2 != 2 || NB16 == 4, "the fused max-pool is built for 64-cout workgroups");
        rows_pool2<R>(p, acc, biasv, smem, b, y0, x0, ct, wave, px, q, tid);
        STAMP_AT(15);
        return;
    }
    if constexpr (FUSE2 == 1) {
        static_assert(FUSE2 != 1 || NB16 == 4, "the fused RGB tail needs all 64 channels of a pixel in one workgroup");
        rows_fuse2<R>(p, acc, biasv, smem, ((int64_t)b * p.tilesY + ty) * tilesX + tx, y0, x0, wave, lane, px, q, tid);
        STAMP_AT(15);
        return;
    }
    if constexpr (NB16 == 4) {
        if (p.pjw) {                                                         // uniform; the host sends such a launch with at most skip 1 and no LDS skip
            if (p.s1) rows_epilogue_proj<R, true>(p, acc, biasv, b, y0, x0, ct, wave, lane, px, q);
            else rows_epilogue_proj<R, false>(p, acc, biasv, b, y0, x0, ct, wave, lane, px, q);
            STAMP_AT(15);
            return;
        }
    }
    if constexpr (SKIP_LDS_OK) if (skip_lds) {
        ConvParams pe = p;                                                  // the epilogue sees only the other skip, as skip 1
        if (p.skip_lds == 1) { pe.s1 = p.s2; pe.s1_cs = p.s2_cs; pe.s1_coff = p.s2_coff; pe.s1_ps = p.s2_ps; pe.s1_rs = p.s2_rs; pe.beta1 = p.beta2; }
        pe.s2 = nullptr;
        rows_epilogue<NB16, R>(pe, acc, biasv, b, y0, x0, ct, wave, px, q);
        STAMP_AT(15);
        return;
    }
    rows_epilogue<NB16, R>(p, acc, biasv, b, y0, x0, ct, wave, px, q);
    STAMP_AT(15);
}

// rows per wave: 16 x 16 output tiles, 12 x 16 for 64 couts per workgroup (see the occupancy note in the header);
// 48/96/192-pixel patches tile exactly either way
template <int NB16, int R = (NB16 == 4 ? 3 : 4)>
int launch_rows(sr_ctx* ctx, const ConvParams& p0, int nct, hipStream_t st) {
    constexpr int lds = (4 * R + 2) * 18 * 64 + 9 * NB16 * 1024 + NB16 * 64;
    static_assert(NB16 != 4 || 4 * R * 16 * 36 * 4 <= lds, "rows_fuse2 parks its tap channels in the tile's LDS");
    ConvParams p = p0;
    const int tilesX = (p.W + 15) / 16;
    p.tilesY = (p.H + 4 * R - 1) / (4 * R);
    if (tilesX > 0xffff || nct > 0x7fff) return ctx->fail(SR_ERR_INVALID, "conv_rows: image too wide / too many cout tiles");
    const int64_t nwg = (int64_t)tilesX * p.tilesY * p.B * nct;
    if (nwg >= (1ll << 31)) return ctx->fail(SR_ERR_INVALID, "conv_rows: too many workgroups for one launch");
    p.tilesX = (nct << 16) | tilesX;
    dim3 grid((unsigned)nwg, 1u);
    if (p.dbg && !p.f2w && !p.plout) {   // diagnostic stamped variant: 16 stamps per workgroup, refused when the buffer is too small for this grid
        if (nwg * 16 * (int64_t)sizeof(unsigned long long) > ctx->stamp_cap)
            return ctx->fail(SR_ERR_INVALID, "conv_rows: the stamp buffer is too small for this launch (sr_debug_stamp_bytes_needed(0, workgroups))");
        auto kd = conv3_rows_kernel<NB16, R, true>;
        if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kd), lds)) return rc;
        hipLaunchKernelGGL(kd, grid, dim3(256), lds, st, p);
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    if constexpr (NB16 == 4 && R == 3) {                     // (the finishing pass assumes 12 x 16 tiles)
        if (p.f2w) {
            if (nct != 1) return ctx->fail(SR_ERR_INVALID, "conv_rows: the fused RGB tail needs a 64-cout conv");
            auto kf = conv3_rows_kernel<NB16, R, false, 1>;
            if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kf), lds)) return rc;
            hipLaunchKernelGGL(kf, grid, dim3(256), lds, st, p);
            SR_HIP(ctx, hipGetLastError());
            return SR_OK;
        }
    }
    if (p.f2w) return ctx->fail(SR_ERR_INVALID, "conv_rows: the fused RGB tail needs a 64-cout conv");
    if constexpr (NB16 == 4 && R == 3) {
        if (p.plout) {
            static_assert(NB16 != 4 || 12 * 16 * 144 <= lds, "rows_pool2 parks the tile's outputs in its LDS");
            auto kp = conv3_rows_kernel<NB16, R, false, 2>;
            if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kp), lds)) return rc;
            hipLaunchKernelGGL(kp, grid, dim3(256), lds, st, p);
            SR_HIP(ctx, hipGetLastError());
            return SR_OK;
        }
    }
    if (p.plout) return ctx->fail(SR_ERR_INVALID, "conv_rows: the fused max-pool needs 64-cout workgroups");
    auto kern = conv3_rows_kernel<NB16, R, false>;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

}  // namespace

// ---- fused RGB tail: host side and the finishing pass ------------------------------------------------------------------------------
namespace {

constexpr int F2_TH = 12, F2_TW = 16, F2_REGION = (F2_TH + 2) * (F2_TW + 2);    // tile of the 64-cout variant (R = 3) and its halo'd region

uint16_t bf16_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// one thread per output pixel: its own tile's partial sum, then the neighbours whose halo'd region covers it (a pixel in a tile's first /
// last row or column is also a halo position of the tile above / below / beside, and of the diagonal one in a corner), always in this order
__global__ void __launch_bounds__(256) rgbtail_finish_kernel(const float* __restrict__ part, int64_t npix, int H, int W, int tilesY, int tilesX,
                                                             const float* __restrict__ bias, int c2, int act, float alpha, int clip, char* out,
                                                             int64_t out_cs, int out_coff, int out_f32) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int x = (int)(i % W);
    const int64_t row = i / W;
    const int y = (int)(row % H);
    const int64_t b = row / H;
    const int ty = y / F2_TH, ry = y - ty * F2_TH + 1, tx = x / F2_TW, rx = x - tx * F2_TW + 1;
    float s[3] = {0.f, 0.f, 0.f};
    auto add = [&](int ty_, int tx_, int ry_, int rx_) {
        const float* q = part + (((b * tilesY + ty_) * tilesX + tx_) * F2_REGION + ry_ * (F2_TW + 2) + rx_) * 3;
        s[0] += q[0]; s[1] += q[1]; s[2] += q[2];
    };
    add(ty, tx, ry, rx);
    const int vy = ry == 1 && ty > 0 ? -1 : (ry == F2_TH && ty + 1 < tilesY ? 1 : 0);
    const int vx = rx == 1 && tx > 0 ? -1 : (rx == F2_TW && tx + 1 < tilesX ? 1 : 0);
    const int nry = vy < 0 ? F2_TH + 1 : 0, nrx = vx < 0 ? F2_TW + 1 : 0;
    if (vy) add(ty + vy, tx, nry, rx);
    if (vx) add(ty, tx + vx, ry, nrx);
    if (vy && vx) add(ty + vy, tx + vx, nry, nrx);
    for (int co = 0; co < c2; ++co) {
        float v = convk::act_apply(s[co] + bias[co], act) * alpha;
        if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
        const int64_t e = i * out_cs + out_coff + co;
        if (out_f32) reinterpret_cast<float*>(out)[e] = v;
        else reinterpret_cast<bf16_t*>(out)[e] = (bf16_t)v;
    }
}

}  // namespace

// A fragment [tb][half]: lane l (tap channel i = l & 15 of block tb, k-quarter q = l >> 4), element j: W2'[t = 16 tb + i][channel 16 (2 half + (j >> 2)) + 4 q + (j & 3)]
// with t = (ky * 3 + kx) * c2 + co -- the channel order rows_fuse2's B fragments have (the accumulator layout of the 64-cout epilogue)
int rgbtail_pack_weights(sr_ctx* ctx, const float* w2, const float* bias2, int c2, RgbTailWeights* out) {
    if (c2 < 1 || c2 > 3) return ctx->fail(SR_ERR_INVALID, "fused RGB tail: 1..3 output channels");
    std::vector<uint16_t> host(4 * 512, 0);
    for (int tb = 0; tb < 2; ++tb)
        for (int hf = 0; hf < 2; ++hf)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int t = 16 * tb + (l & 15);
                    if (t >= 9 * c2) continue;
                    const int tap = t / c2, co = t - tap * c2;
                    const int c = 16 * (2 * hf + (j >> 2)) + 4 * (l >> 4) + (j & 3);
                    host[((tb * 2 + hf) * 64 + l) * 8 + j] = bf16_bits(w2[((size_t)tap * 64 + c) * c2 + co]);
                }
    RgbTailWeights w;
    w.c2 = c2;
    w.a = ctx->dalloc(host.size() * 2);
    if (!w.a) return SR_ERR_OOM;
    w.bias = static_cast<float*>(ctx->dalloc(3 * sizeof(float)));
    if (!w.bias) { ctx->dfree(w.a); return SR_ERR_OOM; }
    float hb[3] = {0.f, 0.f, 0.f};
    if (bias2) for (int i = 0; i < c2; ++i) hb[i] = bias2[i];
    SR_HIP(ctx, hipMemcpy(w.a, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    SR_HIP(ctx, hipMemcpy(w.bias, hb, sizeof hb, hipMemcpyHostToDevice));
    *out = w;
    return SR_OK;
}

// A fragment [nb][half]: lane l (cout i = l & 15 of block nb, k-quarter q = l >> 4), element j: W[channel 16 (2 half + (j >> 2)) + 4 q + (j & 3)][16 nb + i]
int proj_pack_weights(sr_ctx* ctx, const float* w, const float* bias, int cout, ProjWeights* out) {
    if (cout < 16 || cout > 48 || cout % 16 != 0) return ctx->fail(SR_ERR_INVALID, "fused projection: 16, 32 or 48 output channels");
    const int nblk = cout / 16;
    std::vector<uint16_t> host((size_t)nblk * 2 * 512, 0);
    for (int nb = 0; nb < nblk; ++nb)
        for (int hf = 0; hf < 2; ++hf)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * (2 * hf + (j >> 2)) + 4 * (l >> 4) + (j & 3);
                    host[((size_t)(nb * 2 + hf) * 64 + l) * 8 + j] = bf16_bits(w[(size_t)c * cout + 16 * nb + (l & 15)]);
                }
    ProjWeights pw;
    pw.nblk = nblk;
    pw.a = ctx->dalloc(host.size() * 2);
    if (!pw.a) return SR_ERR_OOM;
    pw.bias = static_cast<float*>(ctx->dalloc(sizeof(float) * cout));
    if (!pw.bias) { ctx->dfree(pw.a); return SR_ERR_OOM; }
    std::vector<float> hb(cout, 0.f);
    if (bias) for (int i = 0; i < cout; ++i) hb[i] = bias[i];
    SR_HIP(ctx, hipMemcpy(pw.a, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    SR_HIP(ctx, hipMemcpy(pw.bias, hb.data(), sizeof(float) * cout, hipMemcpyHostToDevice));
    *out = pw;
    return SR_OK;
}

void proj_free_weights(sr_ctx* ctx, ProjWeights* w) {
    if (w->a) ctx->dfree(w->a);
    if (w->bias) ctx->dfree(w->bias);
    w->a = nullptr; w->bias = nullptr;
}

void rgbtail_free_weights(sr_ctx* ctx, RgbTailWeights* w) {
    if (w->a) ctx->dfree(w->a);
    if (w->bias) ctx->dfree(w->bias);
    w->a = nullptr; w->bias = nullptr;
}

int64_t rgbtail_partial_bytes(int B, int H, int W) {
    return (int64_t)B * ((H + F2_TH - 1) / F2_TH) * ((W + F2_TW - 1) / F2_TW) * F2_REGION * 3 * (int64_t)sizeof(float);
}

int rgbtail_finish_launch(sr_ctx* ctx, const RgbTailWeights& w, const float* part, int B, int H, int W, int act, float alpha, int clip01, void* y,
                          int64_t y_cs, int y_coff, int out_f32, hipStream_t st) {
    const int64_t npix = (int64_t)B * H * W;
    if (npix <= 0 || !part || !y || !w.a) return ctx->fail(SR_ERR_INVALID, "fused RGB tail: bad arguments");
    const int64_t nwg = (npix + 255) / 256;
    if (nwg >= (1ll << 31)) return ctx->fail(SR_ERR_INVALID, "fused RGB tail: too many pixels for one launch");
    int rec = -1;
    if (ctx->prof) rec = ctx->prof_open("rgbtail_finish", 0.0, (double)rgbtail_partial_bytes(B, H, W) + (double)npix * w.c2 * (out_f32 ? 4 : 2), st);
    hipLaunchKernelGGL(rgbtail_finish_kernel, dim3((unsigned)nwg), dim3(256), 0, st, part, npix, H, W, (H + F2_TH - 1) / F2_TH, (W + F2_TW - 1) / F2_TW,
                       w.bias, w.c2, act, alpha, clip01, static_cast<char*>(y), y_cs, y_coff, out_f32);
    ctx->prof_close(rec, st);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int conv_rows_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p, hipStream_t st) {
    const int nct = w.CoutP / 16 / w.NT;   // NT holds NB16 for this variant
    switch (w.NT) {
        case 1: return launch_rows<1>(ctx, p, nct, st);
        case 2: return launch_rows<2>(ctx, p, nct, st);
        case 4: return launch_rows<4>(ctx, p, nct, st);   // (round 3, measured again for the 64-channel layers of the output end: 24 x 16 tiles at two workgroups per CU, 256 registers: 3-5 % slower)
    }
    return ctx->fail(SR_ERR_INVALID, "conv_rows: unsupported cout block count");
}
