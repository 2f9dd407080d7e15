// conv_common.h -- pieces shared by the conv kernels: launch parameters and the fused epilogue.
#pragma once
#include "common.h"

namespace convk {

// A tensor view is addressed as  base + (b*H + y)*rs + x*cs + choff(coff + c),  choff(c) = (c >> 5)*ps + (c & 31)  (elements):
//   NHWC                : cs = channels of the buffer, ps = 32 (so choff(c) = c), rs = W*cs;
//   row-blocked (blk)   : [B][H][C/32][W][32] -- cs = 32, ps = W*32, rs = W*C.  A 32-channel block of an image row is contiguous,
//                         so the 3x3 kernel's per-chunk reads and its 32-cout writes move whole 128-byte lines
//                         (only conv_rows, with its fast epilogue, touches such views: conv_launch checks).
struct ConvParams {
    const char* in; int64_t in_cs; int in_coff; int in_ps; int in_rs;
    const char* w; const float* bias;
    char* out; int64_t out_cs; int out_coff; int out_ps; int out_rs; int out_f32;
    const char* s1; int64_t s1_cs; int s1_coff; int s1_ps; int s1_rs; float beta1;
    const char* s2; int64_t s2_cs; int s2_coff; int s2_ps; int s2_rs; float beta2;
    float alpha; int act; int clip; int r; int Cd;
    int B, H, W, Cout;
    int nchunks;        // wide: Cin chunks; thin: number of k-groups (taps pairs)
    int tilesX, tilesY;
    int vec;            // epilogue may use 4-element vector loads/stores
    int skip_lds;       // conv_rows: skip 1 / 2 (value 1 / 2) is input channels [0, Cout) of this conv -- folded in from LDS
    float skip_scale;   // its beta / alpha
    unsigned long long* dbg;   // diagnostic builds only: 16 s_memtime stamps per workgroup (sr_debug_set_stamp_buffer)
    // conv_rows, fused RGB tail (conv_rows.hip, rows_fuse2): this conv's output is consumed on chip by the following 3x3 conv to f2c <= 3
    // channels; the workgroup writes that conv's partial sums over its halo'd tile instead of its own 64-channel output
    const char* f2w;    // the second conv's kernel as four 1 KiB MFMA A-fragments [t block][channel half] (rgbtail_pack_weights), or nullptr
    float* f2part;      // [B][tilesY][tilesX][(TH + 2) * 18][3] fp32 partial sums
    int f2c;            // the second conv's output channels
    // conv_rows, fused 1x1 projection (conv_rows.hip, rows_epilogue_proj): beside its own 64-channel output the conv writes a 1x1 conv of
    // that output (the f / g / h projections of the SelfAttention layer that follows, ESRGAN_model.py:48-56) into a second NHWC buffer
    const char* pjw;    // [cout block < pj_nblk][channel half] 1 KiB MFMA A-fragments (proj_pack_weights), or nullptr
    const float* pjbias;
    char* pjout; int64_t pj_cs; int pj_coff; int pj_rs; int pj_nblk;
    // fp32 thin kernel, fused 1x1 (conv.hip): the conv's activation feeds a 1x1 conv to <= 32 channels from the accumulators; only that is stored
    const float* pw2w; const float* pw2bias; int pw2_cout, pw2_act;
    int splitk_ok;      // fp32 3x3 wide kernel: the split-K variant may be chosen (training entry points only)
    int cell_h, cell_w; // conv_rows fast epilogue: separator rows / columns of a CellGrid layout are not stored (0 = none)
    // conv_rows, fused 2x2 max-pool (conv_rows.hip, rows_pool2): the conv's output is not stored; its VALID 2x2 / stride-2 maximum goes to plout
    // ([B][H/2][W/2] NHWC pixels of pl_cs channels, or packed in a CellGrid: pl_gx images per row of cells of pl_ch x pl_cw pixels, pl_Wv pixels per row)
    char* plout; int64_t pl_cs; int pl_coff; int pl_gx, pl_ch, pl_cw, pl_Wv;
};

__device__ __forceinline__ int choff(int c, int ps) { return (c >> 5) * ps + (c & 31); }
__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case SR_ACT_RELU: return fmaxf(v, 0.f);
        case SR_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case SR_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

template <typename T> __device__ __forceinline__ void load4(const char* base, int64_t eoff, bool vec, int n, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const char* base, int64_t eoff, bool vec, int n, float v[4]) {
    const float* p = reinterpret_cast<const float*>(base) + eoff;
    if (vec) { f32x4 t = *reinterpret_cast<const f32x4*>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else { for (int e = 0; e < 4; ++e) v[e] = e < n ? p[e] : 0.f; }
}
template <> __device__ __forceinline__ void load4<bf16_t>(const char* base, int64_t eoff, bool vec, int n, float v[4]) {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(base) + eoff;
    if (vec) { bf16x4 t = *reinterpret_cast<const bf16x4*>(p); for (int e = 0; e < 4; ++e) v[e] = (float)t[e]; }
    else { for (int e = 0; e < 4; ++e) v[e] = e < n ? (float)p[e] : 0.f; }
}
template <typename T> __device__ __forceinline__ void store4(char* base, int64_t eoff, bool vec, int n, const float v[4], bool f32);
template <> __device__ __forceinline__ void store4<float>(char* base, int64_t eoff, bool vec, int n, const float v[4], bool) {
    float* p = reinterpret_cast<float*>(base) + eoff;
    if (vec) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(p) = t; }
    else { for (int e = 0; e < n; ++e) p[e] = v[e]; }
}
template <> __device__ __forceinline__ void store4<bf16_t>(char* base, int64_t eoff, bool vec, int n, const float v[4], bool f32) {
    if (f32) { store4<float>(base, eoff, vec, n, v, true); return; }
    bf16_t* p = reinterpret_cast<bf16_t*>(base) + eoff;
    if (vec) { bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; *reinterpret_cast<bf16x4*>(p) = t; }
    else { for (int e = 0; e < n; ++e) p[e] = (bf16_t)v[e]; }
}


// Fused epilogue for one output pixel (b, oy, ox) and the 4 consecutive output channels starting at c0:
// bias, activation, alpha, two scaled skips, clip[0,1], then an NHWC store or a depth_to_space (TF "DCR") store.  NHWC views
// only: the host lets row-blocked views reach nothing but conv_rows' fast epilogue (conv_launch).
template <typename T>
__device__ __forceinline__ void epilogue4(const ConvParams& p, int b, int oy, int ox, int c0, const float a[4]) {
    if (c0 >= p.Cout) return;
    const bool vec = p.vec != 0;
    const int64_t pix = ((int64_t)b * p.H + oy) * p.W + ox;
    const int nv = min(4, p.Cout - c0);
    const bool v4 = vec && nv == 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(a[e] + p.bias[c0 + e], p.act) * p.alpha;
    if (p.s1) {
        float s[4];
        load4<T>(p.s1, pix * p.s1_cs + p.s1_coff + c0, v4, nv, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p.beta1 * s[e];
    }
    if (p.s2) {
        float s[4];
        load4<T>(p.s2, pix * p.s2_cs + p.s2_coff + c0, v4, nv, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p.beta2 * s[e];
    }
    if (p.clip) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 1.f);
    }
    if (p.r <= 1) {
        store4<T>(p.out, pix * p.out_cs + p.out_coff + c0, v4, nv, v, p.out_f32 != 0);
    } else if (v4) {   // Cd % 4 == 0 guaranteed by the host when vec
        const int sub = c0 / p.Cd, c = c0 - sub * p.Cd;
        const int i = sub / p.r, j = sub - i * p.r;
        const int64_t dst = (((int64_t)b * p.H * p.r + (int64_t)oy * p.r + i) * ((int64_t)p.W * p.r) + (int64_t)ox * p.r + j) * p.out_cs + p.out_coff + c;
        store4<T>(p.out, dst, true, 4, v, p.out_f32 != 0);
    } else {
        for (int e = 0; e < nv; ++e) {   // TF depth_to_space "DCR": cout = (i*r + j)*Cd + c
            const int co = c0 + e, sub = co / p.Cd, c = co - sub * p.Cd;
            const int i = sub / p.r, j = sub - i * p.r;
            const int64_t dst = (((int64_t)b * p.H * p.r + (int64_t)oy * p.r + i) * ((int64_t)p.W * p.r) + (int64_t)ox * p.r + j) * p.out_cs + p.out_coff + c;
            store4<T>(p.out, dst, false, 1, v + e, p.out_f32 != 0);
        }
    }
}

}  // namespace convk

// bf16 3x3 "row-sliding" kernel (conv_rows.hip)
int conv_rows_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p, hipStream_t st);
// bf16 3x3 from 64 input channels, 64 couts per workgroup, as a persistent kernel with resident weights (conv_stream.hip)
bool conv_stream_supported(const ConvWeights& w, const convk::ConvParams& p);
int conv_stream_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p, hipStream_t st);
// bf16 1x1 streaming kernel (conv_pw.hip)
int conv_pw_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p, hipStream_t st);
