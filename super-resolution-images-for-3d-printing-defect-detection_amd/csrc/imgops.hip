// imgops.hip -- the HBM-bound kernels either side of the conv stacks: dtype/channel padding,
// bicubic resize (OpenCV INTER_CUBIC semantics), PSNR / SSIM (tf.image semantics), patch
// extraction with reflect padding, overlap-add reconstruction, max-pool, global average pool,
// dense head + softmax.  Each is priced against HBM bytes in DESIGN.md; none is GEMM-shaped.
#include "common.h"
#include <algorithm>

namespace {

template <typename T> __device__ __forceinline__ float ldf(const void* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const void* p, int64_t i) { return (float)static_cast<const bf16_t*>(p)[i]; }
template <> __device__ __forceinline__ float ldf<uint8_t>(const void* p, int64_t i) { return (float)static_cast<const uint8_t*>(p)[i]; }
template <typename T> __device__ __forceinline__ void stf(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(void* p, int64_t i, float v) { static_cast<float*>(p)[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(void* p, int64_t i, float v) { static_cast<bf16_t*>(p)[i] = (bf16_t)v; }

__device__ __forceinline__ float ld_dt(const void* p, int64_t i, int dt) {
    return dt == SR_DTYPE_F32 ? ldf<float>(p, i) : (dt == SR_DTYPE_BF16 ? ldf<bf16_t>(p, i) : ldf<uint8_t>(p, i));
}
__device__ __forceinline__ void st_dt(void* p, int64_t i, float v, int dt) {
    if (dt == SR_DTYPE_F32) stf<float>(p, i, v); else stf<bf16_t>(p, i, v);
}

// ---------------------------------------------------------------------------------- convert + pad
__global__ void convert_pad_kernel(const void* x, int in_dt, int64_t npix, int C, void* y, int out_dt, int Cp, float mul, float add) {
    const int64_t n = npix * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / Cp;
        const int c = (int)(i - pix * Cp);
        const float v = c < C ? ld_dt(x, pix * C + c, in_dt) * mul + add : 0.f;
        st_dt(y, i, v, out_dt);
    }
}

// ---------------------------------------------------------------------------------- maxpool 2x2 (VALID, floor)
// pixel index of (image b, row y, column x) of a batch of H x W images, plain or packed in a CellGrid (common.h); `rw` = pixels per buffer row
__device__ __forceinline__ int64_t grid_pixel(const CellGrid& g, int64_t b, int y, int x, int H, int W, int* rw) {
    if (g.gx == 0) { *rw = W; return (b * H + y) * W + x; }
    *rw = g.Wv;
    return ((b / g.gx) * g.ch + y) * (int64_t)g.Wv + (b % g.gx) * g.cw + x;
}

__global__ void maxpool2_kernel(const void* x, int dt, int B, int H, int W, int C, int64_t x_cs, void* y, int64_t y_cs, CellGrid gi, CellGrid go) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        int rwi, rwo;
        const int64_t base = grid_pixel(gi, b, 2 * oy, 2 * ox, H, W, &rwi) * x_cs + c;
        const float v = fmaxf(fmaxf(ld_dt(x, base, dt), ld_dt(x, base + x_cs, dt)),
                              fmaxf(ld_dt(x, base + (int64_t)rwi * x_cs, dt), ld_dt(x, base + (int64_t)rwi * x_cs + x_cs, dt)));
        st_dt(y, grid_pixel(go, b, oy, ox, Ho, Wo, &rwo) * y_cs + c, v, dt);
    }
}

// bf16, 8 channels (16 B) per thread: the VGG16 pools move 2.9 GB per 1024 patches of 96 x 96; element by element that ran at ~1.5 TB/s.
// max of bf16 values is exact in any order, so the result is the scalar kernel's bit for bit.
__global__ void __launch_bounds__(256) maxpool2_bf16x8_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C8, int64_t x_cs, bf16_t* __restrict__ y,
                                                              int64_t y_cs, CellGrid gi, CellGrid go) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * Ho * Wo * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C8) * 8; t /= C8;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int64_t b = t / Ho;
        int rwi, rwo;
        const bf16_t* s = x + grid_pixel(gi, b, 2 * oy, 2 * ox, H, W, &rwi) * x_cs + c;
        const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(s), v01 = *reinterpret_cast<const bf16x8*>(s + x_cs);
        const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(s + (int64_t)rwi * x_cs), v11 = *reinterpret_cast<const bf16x8*>(s + (int64_t)rwi * x_cs + x_cs);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(fmaxf((float)v00[e], (float)v01[e]), fmaxf((float)v10[e], (float)v11[e]));
        *reinterpret_cast<bf16x8*>(y + grid_pixel(go, b, oy, ox, Ho, Wo, &rwo) * y_cs + c) = o;
    }
}

// ---------------------------------------------------------------------------------- global average pool -> fp32
__global__ void gap_kernel(const void* x, int dt, int HW, int C, int64_t x_cs, float* y) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < HW; ++i) s += ld_dt(x, ((int64_t)b * HW + i) * x_cs + c, dt);
        y[(int64_t)b * C + c] = s / (float)HW;
    }
}

// ---------------------------------------------------------------------------------- dense (+relu | softmax)
__global__ void dense_kernel(const float* x, const float* w, const float* bias, int In, int Out, int act, float* y, int y_dt,
                             void* y_typed) {
    extern __shared__ float sm[];   // [Out] logits + [256] scratch
    float* logits = sm;
    float* red = sm + Out;
    const int b = blockIdx.x;
    const float* xb = x + (int64_t)b * In;
    for (int o = threadIdx.x; o < Out; o += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < In; ++i) s += xb[i] * w[(int64_t)i * Out + o];
        s += bias ? bias[o] : 0.f;
        if (act == SR_ACT_RELU) s = fmaxf(s, 0.f);
        else if (act == SR_ACT_LRELU) s = s > 0.f ? s : 0.2f * s;
        else if (act == 101) s = 1.f / (1.f + expf(-s));             // sigmoid (discriminator output, ESRGAN_model.py:373)
        logits[o] = s;
    }
    __syncthreads();
    if (act == 100) {
        float mx = -INFINITY;
        for (int o = threadIdx.x; o < Out; o += blockDim.x) mx = fmaxf(mx, logits[o]);
        red[threadIdx.x] = mx;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
        mx = red[0];
        __syncthreads();
        float sum = 0.f;
        for (int o = threadIdx.x; o < Out; o += blockDim.x) { const float e = expf(logits[o] - mx); logits[o] = e; sum += e; }
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int s = blockDim.x / 2; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
        sum = red[0];
        for (int o = threadIdx.x; o < Out; o += blockDim.x) logits[o] /= sum;
        __syncthreads();
    }
    for (int o = threadIdx.x; o < Out; o += blockDim.x) {
        if (y) y[(int64_t)b * Out + o] = logits[o];
        if (y_typed) st_dt(y_typed, (int64_t)b * Out + o, logits[o], y_dt);
    }
}

// ---------------------------------------------------------------------------------- bicubic (OpenCV INTER_CUBIC)
// reference: cv2.resize float path (classic_algorithms.py:11-13, SRCNN_model.py:191): half-pixel
// centres, Keys a=-0.75 (three polynomials + 1-sum), replicate border, horizontal then vertical.
__device__ __forceinline__ void cubic_w(float x, float w[4]) {
    const float A = -0.75f;
    w[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    w[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}
__device__ __forceinline__ void cubic_axis(int d, double scale, int n, int idx[4], float w[4]) {
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    cubic_w(f - (float)s, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) idx[k] = min(max(s - 1 + k, 0), n - 1);
}

__global__ void bicubic_f32_kernel(const float* x, int B, int H, int W, int C, int oH, int oW, double sy, double sx, void* y,
                                   int out_dt, int64_t y_cs) {
    const int64_t n = (int64_t)B * oH * oW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % oW);
        const int oy = (int)((i / oW) % oH);
        const int b = (int)(i / ((int64_t)oW * oH));
        int ix[4], iy[4];
        float wx[4], wy[4];
        cubic_axis(ox, sx, W, ix, wx);
        cubic_axis(oy, sy, H, iy, wy);
        const float* xb = x + (int64_t)b * H * W * C;
        for (int c = 0; c < C; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const float* row = xb + (int64_t)iy[ky] * W * C + c;
                float hsum = 0.f;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) hsum += row[(int64_t)ix[kx] * C] * wx[kx];
                acc += hsum * wy[ky];
            }
            st_dt(y, i * y_cs + c, acc, out_dt);
        }
        for (int c = C; c < y_cs; ++c) st_dt(y, i * y_cs + c, 0.f, out_dt);   // zero the pad channels of a padded view
    }
}

// uint8 fixed-point path: 11-bit coefficients, int accumulation, rounding >> 22, saturate.
// Tiled variant for fp32 -> fp32 with a dense output (y_cs == C).  A block produces BT_Y rows x BT_X columns from an LDS copy of
// the source window they touch; a thread owns one output column: its horizontal 4-tap sums over the window rows are computed
// once (cv2 is separable: horizontal then vertical, so sharing them across output rows is the same arithmetic in the same order)
// and parked in LDS, then every output row of the tile is 4 vertical taps per channel.  Rows go back to memory as consecutive
// floats.  The per-pixel kernel above issues 16*C scattered 4-byte global loads and ~250 instructions per output pixel (index
// arithmetic included) and is instruction-bound at 0.7 TB/s of its 4x up-scale stream.  The host picks this kernel when the
// window and the row sums fit the LDS arrays.
constexpr int BT_X = 256, BT_Y = 8, BT_WIN = 6144, BT_ROWS = 12;
__global__ void __launch_bounds__(256) bicubic_f32_tile_kernel(const float* x, int H, int W, int C, int oH, int oW, double sy, double sx, float* y,
                                                               int win_floats) {
    extern __shared__ float bt_smem[];
    float* win = bt_smem;                               // source window [ny][nx][C] (win_floats, host-sized upper bound)
    float* hs = bt_smem + win_floats;                   // horizontal sums [ny][BT_X][C]; later the output tile [BT_Y][BT_X*C]
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * BT_X, oy0 = blockIdx.y * BT_Y, b = blockIdx.z;
    const int oxl = min(ox0 + BT_X, oW) - 1, oyl = min(oy0 + BT_Y, oH) - 1;       // last pixel of the tile
    int ia[4], ib[4];
    float wdummy[4];
    cubic_axis(ox0, sx, W, ia, wdummy); cubic_axis(oxl, sx, W, ib, wdummy);
    const int ix0 = ia[0], nx = ib[3] - ia[0] + 1;                                 // clamped indices are monotonic in the output index
    cubic_axis(oy0, sy, H, ia, wdummy); cubic_axis(oyl, sy, H, ib, wdummy);
    const int iy0 = ia[0], ny = ib[3] - ia[0] + 1;
    const float* xb = x + (int64_t)b * H * W * C;
    const int rowlen = nx * C;
    for (int r = 0; r < ny; ++r) {
        const float* src = xb + ((int64_t)(iy0 + r) * W + ix0) * C;
        for (int k = tid; k < rowlen; k += 256) win[r * rowlen + k] = src[k];
    }
    __shared__ int yrow[BT_Y][4];                        // the vertical taps are the same for every column: one thread per row
    __shared__ float ywt[BT_Y][4];
    if (tid < BT_Y) {
        int iy[4];
        float wy[4];
        cubic_axis(min(oy0 + tid, oH - 1), sy, H, iy, wy);
#pragma unroll
        for (int k = 0; k < 4; ++k) { yrow[tid][k] = (iy[k] - iy0) * BT_X; ywt[tid][k] = wy[k]; }
    }
    __syncthreads();
    const int ox = ox0 + tid;
    if (ox < oW) {
        int ix[4];
        float wx[4];
        cubic_axis(ox, sx, W, ix, wx);
        const int o0 = (ix[0] - ix0) * C, o1 = (ix[1] - ix0) * C, o2 = (ix[2] - ix0) * C, o3 = (ix[3] - ix0) * C;
        for (int r = 0; r < ny; ++r) {
            const float* row = win + r * rowlen;
            for (int c = 0; c < C; ++c) {
                float hsum = 0.f;
                hsum += row[o0 + c] * wx[0]; hsum += row[o1 + c] * wx[1]; hsum += row[o2 + c] * wx[2]; hsum += row[o3 + c] * wx[3];
                hs[(r * BT_X + tid) * C + c] = hsum;
            }
        }
    }
    // a column's sums are read back only by the thread that wrote them: no barrier needed before the vertical pass
    float outv[BT_Y][4];
#pragma unroll
    for (int ty = 0; ty < BT_Y; ++ty) {
        const int oy = oy0 + ty;
        if (ox < oW && oy < oH) {
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) acc += hs[(yrow[ty][ky] + tid) * C + c] * ywt[ty][ky];
                outv[ty][c] = acc;
            }
        }
    }
    __syncthreads();                                     // everyone is done with hs: reuse it for the output tile
#pragma unroll
    for (int ty = 0; ty < BT_Y; ++ty)
        for (int c = 0; c < C; ++c) hs[(ty * BT_X + tid) * C + c] = outv[ty][c];
    __syncthreads();
    const int ncols = (min(ox0 + BT_X, oW) - ox0) * C;                             // floats per tile row
#pragma unroll
    for (int r = 0; r < BT_Y; ++r) {
        if (oy0 + r >= oH) break;
        float* dst = y + (((int64_t)b * oH + oy0 + r) * oW + ox0) * C;
        for (int k = tid; k < ncols; k += 256) dst[k] = hs[r * BT_X * C + k];
    }
}

__global__ void bicubic_u8_kernel(const uint8_t* x, int B, int H, int W, int C, int oH, int oW, double sy, double sx, uint8_t* y) {
    const int64_t n = (int64_t)B * oH * oW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % oW);
        const int oy = (int)((i / oW) % oH);
        const int b = (int)(i / ((int64_t)oW * oH));
        int ix[4], iy[4], iwx[4], iwy[4];
        float wx[4], wy[4];
        cubic_axis(ox, sx, W, ix, wx);
        cubic_axis(oy, sy, H, iy, wy);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            iwx[k] = min(max((int)rintf(wx[k] * 2048.f), -32768), 32767);
            iwy[k] = min(max((int)rintf(wy[k] * 2048.f), -32768), 32767);
        }
        const uint8_t* xb = x + (int64_t)b * H * W * C;
        for (int c = 0; c < C; ++c) {
            int acc = 0;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const uint8_t* row = xb + (int64_t)iy[ky] * W * C + c;
                int hsum = 0;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) hsum += (int)row[(int64_t)ix[kx] * C] * iwx[kx];
                acc += hsum * iwy[ky];
            }
            acc = (acc + (1 << 21)) >> 22;
            y[i * C + c] = (uint8_t)min(max(acc, 0), 255);
        }
    }
}

// ---------------------------------------------------------------------------------- reductions
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// stage 1: partial[b][blk] = sum over a slice of (a-b)^2
__global__ void sqdiff_partial_kernel(const float* a, const float* b, int64_t n_per_image, float* partial) {
    __shared__ float red[16];
    const int img = blockIdx.y;
    const float* pa = a + (int64_t)img * n_per_image;
    const float* pb = b + (int64_t)img * n_per_image;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_image; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pa[i] - pb[i];
        s += d * d;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[(int64_t)img * gridDim.x + blockIdx.x] = s;
}
// stage 2 (fixed order -> reproducible): mode 0 = psnr, 1 = plain mean, 2 = ssim mean
__global__ void finish_kernel(const float* partial, int nblk, double count, float max_val, int mode, float* out) {
    __shared__ double red[256];
    const int img = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += (double)partial[(int64_t)img * nblk + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const float mean = (float)(red[0] / count);
        // tf.image.psnr: 20*log(max)/log(10) - 10/log(10)*log(mse)
        out[img] = mode == 0 ? (20.f * logf(max_val) / logf(10.f) - 10.f / logf(10.f) * logf(mean)) : mean;
    }
}

// SSIM (tf.image.ssim: 11x11 gaussian sigma 1.5, VALID, per channel, k1=.01 k2=.03).
struct GaussW { float g[11]; };
constexpr int ST = 32, SWIN = ST + 10;
__global__ void __launch_bounds__(256) ssim_partial_kernel(const float* a, const float* b, int H, int W, int C, GaussW gw, float c1,
                                                           float c2, int tilesX, float* partial) {
    // 32 x 32 outputs per block and channel from a 42 x 42 window.  Both 11-tap passes are register-blocked -- a thread makes 4
    // neighbouring outputs from the 14 inputs they share -- so an output costs ~23 LDS reads instead of ~80 (the 16 x 16,
    // one-output-per-thread version was LDS-bound at 0.7 TB/s of its input stream).  Each output's taps are still summed in
    // order k = 0..10.
    __shared__ float ta[SWIN][SWIN + 1], tb[SWIN][SWIN + 1];
    __shared__ float hz[4][SWIN][ST + 1];
    __shared__ float red[16];
    const int img = blockIdx.y;
    const int ty0 = (blockIdx.x / tilesX) * ST, tx0 = (blockIdx.x % tilesX) * ST;
    const int oH = H - 10, oW = W - 10;
    const int tid = threadIdx.x;
    float g[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) g[k] = gw.g[k];
    float total = 0.f;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
        for (int u = tid; u < SWIN * SWIN; u += 256) {
            const int py = u / SWIN, px = u % SWIN;
            const int gy = min(ty0 + py, H - 1), gx = min(tx0 + px, W - 1);
            const int64_t idx = (((int64_t)img * H + gy) * W + gx) * C + c;
            ta[py][px] = a[idx];
            tb[py][px] = b[idx];
        }
        __syncthreads();
        // horizontal pass: (row, group of 4 columns)
        for (int u = tid; u < SWIN * (ST / 4); u += 256) {
            const int py = u / (ST / 4), px0 = (u % (ST / 4)) * 4;
            float va[14], vb[14], vab[14], vq[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                va[i] = ta[py][px0 + i]; vb[i] = tb[py][px0 + i];
                vab[i] = va[i] * vb[i]; vq[i] = va[i] * va[i] + vb[i] * vb[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) { s0 += g[k] * va[j + k]; s1 += g[k] * vb[j + k]; s2 += g[k] * vab[j + k]; s3 += g[k] * vq[j + k]; }
                hz[0][py][px0 + j] = s0; hz[1][py][px0 + j] = s1; hz[2][py][px0 + j] = s2; hz[3][py][px0 + j] = s3;
            }
        }
        __syncthreads();
        // vertical pass: (column, group of 4 rows)
        {
            const int tx = tid % ST, ry0 = (tid / ST) * 4;
            float m0[4] = {0.f, 0.f, 0.f, 0.f}, m1[4] = {0.f, 0.f, 0.f, 0.f}, sab[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
            float h0[14], h1[14], h2[14], h3[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) { h0[i] = hz[0][ry0 + i][tx]; h1[i] = hz[1][ry0 + i][tx]; h2[i] = hz[2][ry0 + i][tx]; h3[i] = hz[3][ry0 + i][tx]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 11; ++k) { m0[j] += g[k] * h0[j + k]; m1[j] += g[k] * h1[j + k]; sab[j] += g[k] * h2[j + k]; sq[j] += g[k] * h3[j + k]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ty0 + ry0 + j < oH && tx0 + tx < oW) {
                    const float num0 = m0[j] * m1[j] * 2.f, den0 = m0[j] * m0[j] + m1[j] * m1[j];
                    const float lum = (num0 + c1) / (den0 + c1);
                    const float cs = (sab[j] * 2.f - num0 + c2) / (sq[j] - den0 + c2);
                    total += lum * cs;
                }
            }
        }
    }
    total = block_sum(total, red);
    if (tid == 0) partial[(int64_t)img * gridDim.x + blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------- patches
// reflect index for bottom/right padding (np.pad mode='reflect': no edge repeat): y >= n -> 2(n-1) - y
__device__ __forceinline__ int reflect(int y, int n) { return y < n ? y : 2 * (n - 1) - y; }

__global__ void extract_patches_kernel(const float* img, int H, int W, int C, int patch, int stride, float mul, float add, int out_dt,
                                       void* out, int ny, int nx) {
    const int64_t n = (int64_t)ny * nx * patch * patch * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int px = (int)(t % patch); t /= patch;
        const int py = (int)(t % patch); t /= patch;
        const int jx = (int)(t % nx);
        const int jy = (int)(t / nx);
        const int y = reflect(jy * stride + py, H), x = reflect(jx * stride + px, W);
        st_dt(out, i, img[((int64_t)y * W + x) * C + c] * mul + add, out_dt);
    }
}

// gather form of the reference's scatter-add: same summation order (patches in row-major position order)
__global__ void overlap_add_kernel(const void* patches, int in_dt, int H, int W, int C, int patch, int stride, int scale, float mul,
                                   float add, int ny, int nx, float* out) {
    const int oH = H * scale, oW = W * scale, ps = patch * scale, ss = stride * scale;
    const int64_t n = (int64_t)oH * oW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int x = (int)(t % oW);
        const int y = (int)(t / oW);
        const int jy1 = min(y / ss, ny - 1), jx1 = min(x / ss, nx - 1);
        const int jy0 = max(0, (y - ps + ss) / ss), jx0 = max(0, (x - ps + ss) / ss);
        float sum = 0.f, cnt = 0.f;
        for (int jy = jy0; jy <= jy1; ++jy) {
            const int py = y - jy * ss;
            if (py < 0 || py >= ps) continue;
            for (int jx = jx0; jx <= jx1; ++jx) {
                const int px = x - jx * ss;
                if (px < 0 || px >= ps) continue;
                const int64_t idx = ((((int64_t)jy * nx + jx) * ps + py) * ps + px) * C + c;
                sum += ld_dt(patches, idx, in_dt) * mul + add;
                cnt += 1.f;
            }
        }
        const float v = cnt != 0.f ? sum / cnt : 0.f;
        out[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

inline unsigned grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    const int64_t cap = 256 * 8 * 4;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ================================================================================================
// NHWC channels [src_coff, src_coff + C) -> the same channels of a row-blocked buffer [B][H][dst_C/32][W][32] at dst_coff (bf16;
// C, offsets multiples of 32).  One 16-byte unit per thread, destination-linear within a 32-channel row segment.
__global__ void nhwc_to_blocked_kernel(const bf16_t* src, int64_t src_cs, int src_coff, int64_t rows, int W, int C, bf16_t* dst, int64_t dst_C,
                                       int dst_coff) {
    const int64_t n = rows * (C / 32) * W * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i & 3);
        int64_t t = i >> 2;
        const int x = (int)(t % W); t /= W;
        const int cb = (int)(t % (C / 32));
        const int64_t row = t / (C / 32);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (row * W + x) * src_cs + src_coff + cb * 32 + sl * 8);
        *reinterpret_cast<bf16x8*>(dst + row * W * dst_C + (int64_t)(dst_coff / 32 + cb) * W * 32 + x * 32 + sl * 8) = v;
    }
}

int nhwc_to_blocked_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C,
                           int dst_coff, hipStream_t st) {
    if (C % 32 || src_coff % 8 || dst_coff % 32 || dst_C % 32 || src_cs % 8) return ctx->fail(SR_ERR_INVALID, "nhwc_to_blocked: 32-channel granularity");
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(nhwc_to_blocked_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const bf16_t*>(src), src_cs, src_coff,
                       (int64_t)B * H, W, C, static_cast<bf16_t*>(dst), dst_C, dst_coff);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// Two-up packing of 24-pixel-wide images for the fused dense-block kernels (round 4; dense_fused.hip SEAM): image b sits in columns [24 (b & 1), 24 (b & 1) + 24) of
// image b >> 1 of a row-blocked buffer of 48-pixel rows, [ceil(B / 2)][H][dst_C / 32][48][32].  pack: NHWC channels -> packed (an odd batch's missing half is written
// as zeros); unpack: packed -> the ordinary row-blocked layout [B][H][dst_C / 32][W][32].  bf16, 32-channel granularity, one 16-byte unit per thread.
__global__ void pack_pairs_kernel(const bf16_t* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, bf16_t* dst, int64_t dst_C, int dst_coff) {
    const int Bp = (B + 1) & ~1;
    const int64_t n = (int64_t)Bp * H * (C / 32) * W * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i & 3);
        int64_t t = i >> 2;
        const int x = (int)(t % W); t /= W;
        const int cb = (int)(t % (C / 32)); t /= (C / 32);
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        bf16x8 v = {};
        if (b < B) v = *reinterpret_cast<const bf16x8*>(src + (((int64_t)b * H + y) * W + x) * src_cs + src_coff + cb * 32 + sl * 8);
        *reinterpret_cast<bf16x8*>(dst + ((int64_t)(b >> 1) * H + y) * (2 * W) * dst_C + (int64_t)(dst_coff / 32 + cb) * (2 * W) * 32 + ((b & 1) * W + x) * 32 + sl * 8) = v;
    }
}

__global__ void unpack_pairs_kernel(const bf16_t* src, int64_t src_C, int src_coff, int B, int H, int W, int C, bf16_t* dst, int64_t dst_C, int dst_coff) {
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i & 3);
        int64_t t = i >> 2;
        const int x = (int)(t % W); t /= W;
        const int cb = (int)(t % (C / 32)); t /= (C / 32);
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + ((int64_t)(b >> 1) * H + y) * (2 * W) * src_C + (int64_t)(src_coff / 32 + cb) * (2 * W) * 32 + ((b & 1) * W + x) * 32 + sl * 8);
        *reinterpret_cast<bf16x8*>(dst + ((int64_t)b * H + y) * W * dst_C + (int64_t)(dst_coff / 32 + cb) * W * 32 + x * 32 + sl * 8) = v;
    }
}

int pack_pairs_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, hipStream_t st) {
    if (C % 32 || src_coff % 8 || dst_coff % 32 || dst_C % 32 || src_cs % 8) return ctx->fail(SR_ERR_INVALID, "pack_pairs: 32-channel granularity");
    const int64_t n = (int64_t)((B + 1) & ~1) * H * (C / 32) * W * 4;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(pack_pairs_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const bf16_t*>(src), src_cs, src_coff, B, H, W, C, static_cast<bf16_t*>(dst), dst_C, dst_coff);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int unpack_pairs_launch(sr_ctx* ctx, const void* src, int64_t src_C, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, hipStream_t st) {
    if (C % 32 || src_coff % 32 || dst_coff % 32 || dst_C % 32 || src_C % 32) return ctx->fail(SR_ERR_INVALID, "unpack_pairs: 32-channel granularity");
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(unpack_pairs_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const bf16_t*>(src), src_C, src_coff, B, H, W, C, static_cast<bf16_t*>(dst), dst_C, dst_coff);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// Cell packing of small images for the TILE kernels (round 4; api.hip `cellpack`): the batch as ONE image of ceil(B / gx) x gx cells of (H + 1) x (W + 1) pixels, the last row /
// column of a cell a zero separator (CellGrid, common.h), row-blocked: [Hv][dst_C / 32][Wv][32].  24-pixel-wide images fill 56 % of the 16 x 16 output tiles they are cut into,
// the grid 92 %.  pack: NHWC channels -> the grid's image pixels (separators and unused cells are zero from the buffer's adoption and never written); unpack: grid -> the
// ordinary row-blocked layout [B][H][dst_C / 32][W][32].  bf16, 32-channel granularity, one 16-byte unit per thread.
__global__ void cell_pack_kernel(const bf16_t* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, bf16_t* dst, int64_t dst_C, int dst_coff, int gx, int ch, int cw, int Wv) {
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i & 3);
        int64_t t = i >> 2;
        const int x = (int)(t % W); t /= W;
        const int cb = (int)(t % (C / 32)); t /= (C / 32);
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (((int64_t)b * H + y) * W + x) * src_cs + src_coff + cb * 32 + sl * 8);
        const int64_t row = (int64_t)(b / gx) * ch + y;
        const int col = (b % gx) * cw + x;
        *reinterpret_cast<bf16x8*>(dst + row * Wv * dst_C + (int64_t)(dst_coff / 32 + cb) * Wv * 32 + col * 32 + sl * 8) = v;
    }
}

__global__ void cell_unpack_kernel(const bf16_t* src, int64_t src_C, int src_coff, int B, int H, int W, int C, bf16_t* dst, int64_t dst_C, int dst_coff, int gx, int ch, int cw, int Wv) {
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i & 3);
        int64_t t = i >> 2;
        const int x = (int)(t % W); t /= W;
        const int cb = (int)(t % (C / 32)); t /= (C / 32);
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const int64_t row = (int64_t)(b / gx) * ch + y;
        const int col = (b % gx) * cw + x;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + row * Wv * src_C + (int64_t)(src_coff / 32 + cb) * Wv * 32 + col * 32 + sl * 8);
        *reinterpret_cast<bf16x8*>(dst + ((int64_t)b * H + y) * W * dst_C + (int64_t)(dst_coff / 32 + cb) * W * 32 + x * 32 + sl * 8) = v;
    }
}

int cell_pack_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, const CellGrid& g, hipStream_t st) {
    if (C % 32 || src_coff % 8 || dst_coff % 32 || dst_C % 32 || src_cs % 8 || g.gx < 1 || g.ch != H + 1 || g.cw != W + 1) return ctx->fail(SR_ERR_INVALID, "cell_pack: 32-channel granularity, cells of (H + 1) x (W + 1)");
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(cell_pack_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const bf16_t*>(src), src_cs, src_coff, B, H, W, C, static_cast<bf16_t*>(dst), dst_C, dst_coff, g.gx, g.ch, g.cw, g.Wv);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int cell_unpack_launch(sr_ctx* ctx, const void* src, int64_t src_C, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, const CellGrid& g, hipStream_t st) {
    if (C % 32 || src_coff % 32 || dst_coff % 32 || dst_C % 32 || src_C % 32 || g.gx < 1 || g.ch != H + 1 || g.cw != W + 1) return ctx->fail(SR_ERR_INVALID, "cell_unpack: 32-channel granularity, cells of (H + 1) x (W + 1)");
    const int64_t n = (int64_t)B * H * (C / 32) * W * 4;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(cell_unpack_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const bf16_t*>(src), src_C, src_coff, B, H, W, C, static_cast<bf16_t*>(dst), dst_C, dst_coff, g.gx, g.ch, g.cw, g.Wv);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// Diagnostic tap (sr_model_set_tap): channels [coff, coff + C) of an activation buffer -- NHWC or row-blocked, bf16 or fp32 -- as a dense
// fp32 NHWC tensor.  One thread per element; never on a timed path.
__global__ void tap_copy_kernel(const void* src, int dtype, int blk, int64_t cs, int coff, int64_t rows, int W, int C, float* dst, int pair_h) {
    const int64_t n = rows * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        int x = (int)(pix % W);
        int64_t row = pix / W;
        int Wb = W;
        if (pair_h) {                                          // two-up packed buffer (pack_pairs_kernel): image b = row / H rides in half b & 1 of packed image b >> 1
            const int64_t b = row / pair_h;
            row = (b >> 1) * pair_h + row % pair_h;
            x += (int)(b & 1) * W;
            Wb = 2 * W;
        }
        const int ch = coff + c;
        const int64_t e = blk ? row * Wb * cs + (int64_t)(ch >> 5) * Wb * 32 + x * 32 + (ch & 31) : pix * cs + ch;
        dst[i] = dtype == SR_DTYPE_BF16 ? (float)static_cast<const bf16_t*>(src)[e] : static_cast<const float*>(src)[e];
    }
}

int tap_copy_launch(sr_ctx* ctx, const void* src, int dtype, int blk, int64_t cs, int coff, int B, int H, int W, int C, float* dst, hipStream_t st, bool pairs) {
    if (pairs && !blk) return ctx->fail(SR_ERR_STATE, "tap: two-up packing is a row-blocked layout");
    const int64_t n = (int64_t)B * H * W * C;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(tap_copy_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, dtype, blk, cs, coff, (int64_t)B * H, W, C, dst, pairs ? H : 0);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int convert_pad_launch(sr_ctx* ctx, const void* x, int in_dtype, int64_t npix, int C, void* y, int out_dtype, int Cp, float mul,
                       float add, hipStream_t st) {
    if (npix <= 0) return SR_OK;
    hipLaunchKernelGGL(convert_pad_kernel, dim3(grid_for(npix * Cp)), dim3(256), 0, st, x, in_dtype, npix, C, y, out_dtype, Cp, mul, add);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int maxpool2_launch(sr_ctx* ctx, int dtype, const void* x, int B, int H, int W, int C, int64_t x_cs, void* y, int64_t y_cs, hipStream_t st, CellGrid gi,
                    CellGrid go) {
    const int64_t n = (int64_t)B * (H / 2) * (W / 2) * C;
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "maxpool: output would be empty");
    if (dtype == SR_DTYPE_BF16 && C % 8 == 0 && x_cs % 8 == 0 && y_cs % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) {
        hipLaunchKernelGGL(maxpool2_bf16x8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, static_cast<const bf16_t*>(x), B, H, W, C / 8, x_cs,
                           static_cast<bf16_t*>(y), y_cs, gi, go);
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, dtype, B, H, W, C, x_cs, y, y_cs, gi, go);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int gap_launch(sr_ctx* ctx, int dtype, const void* x, int B, int HW, int C, int64_t x_cs, float* y, hipStream_t st) {
    hipLaunchKernelGGL(gap_kernel, dim3(B), dim3(256), 0, st, x, dtype, HW, C, x_cs, y);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int dense_launch(sr_ctx* ctx, const float* x, const float* w, const float* bias, int B, int In, int Out, int act, float* y, int y_dtype,
                 void* y_typed, hipStream_t st) {
    if (Out > 8192) return ctx->fail(SR_ERR_INVALID, "dense: Out too large");
    hipLaunchKernelGGL(dense_kernel, dim3(B), dim3(256), sizeof(float) * (Out + 256), st, x, w, bias, In, Out, act, y, y_dtype, y_typed);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int bicubic_launch(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW, void* y, int out_dtype,
                   int64_t y_cs, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || outH <= 0 || outW <= 0) return ctx->fail(SR_ERR_INVALID, "bicubic: empty tensor");
    // OpenCV: inv_scale = dsize/ssize (double), scale = 1./inv_scale
    const double sy = 1.0 / ((double)outH / (double)H), sx = 1.0 / ((double)outW / (double)W);
    const int64_t n = (int64_t)B * outH * outW;
    // source window of a BT_Y x BT_X output tile: (tile extent * scale + 4 taps + 1) per axis
    const int64_t win_rows = (int)(BT_Y * sy) + 6, win = (int64_t)((int)(BT_X * sx) + 6) * win_rows * C;
    if (dtype == SR_DTYPE_F32 && out_dtype == SR_DTYPE_F32 && y_cs == C && C <= 4 && win <= BT_WIN && win_rows <= BT_ROWS && B <= 65535 &&
        (outH + BT_Y - 1) / BT_Y <= 65535)
        hipLaunchKernelGGL(bicubic_f32_tile_kernel, dim3((outW + BT_X - 1) / BT_X, (outH + BT_Y - 1) / BT_Y, B), dim3(256),
                           (size_t)(win + (win_rows > BT_Y ? win_rows : BT_Y) * BT_X * C) * sizeof(float), st,
                           static_cast<const float*>(x), H, W, C, outH, outW, sy, sx, static_cast<float*>(y), (int)win);
    else if (dtype == SR_DTYPE_F32)
        hipLaunchKernelGGL(bicubic_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const float*>(x), B, H, W, C, outH, outW,
                           sy, sx, y, out_dtype, y_cs);
    else if (dtype == SR_DTYPE_U8)
        hipLaunchKernelGGL(bicubic_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const uint8_t*>(x), B, H, W, C, outH, outW,
                           sy, sx, static_cast<uint8_t*>(y));
    else return ctx->fail(SR_ERR_INVALID, "bicubic: dtype must be f32 or u8");
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

static int reduce_scratch(sr_ctx* ctx, size_t floats, float** out) {
    void* p = ctx->scratch(floats * sizeof(float));   // ctx-owned, grown on demand, reused by later calls on the same stream
    if (!p) return SR_ERR_OOM;
    *out = static_cast<float*>(p);
    return SR_OK;
}

int psnr_launch(sr_ctx* ctx, const float* a, const float* b, int B, int64_t n_per_image, float max_val, float* out, hipStream_t st) {
    if (B <= 0 || n_per_image <= 0) return ctx->fail(SR_ERR_INVALID, "psnr: empty tensor");
    const int nblk = (int)grid_for(n_per_image) > 1024 ? 1024 : (int)grid_for(n_per_image);
    float* partial;
    int rc = reduce_scratch(ctx, (size_t)B * nblk, &partial);
    if (rc) return rc;
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(nblk, B), dim3(256), 0, st, a, b, n_per_image, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(B), dim3(256), 0, st, partial, nblk, (double)n_per_image, max_val, 0, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int mse_launch(sr_ctx* ctx, const float* a, const float* b, int64_t n, float* out, hipStream_t st) {
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "mse: empty tensor");
    const int nblk = (int)grid_for(n) > 1024 ? 1024 : (int)grid_for(n);
    float* partial;
    int rc = reduce_scratch(ctx, (size_t)nblk, &partial);
    if (rc) return rc;
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(nblk, 1), dim3(256), 0, st, a, b, n, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, partial, nblk, (double)n, 1.f, 1, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int ssim_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float max_val, float* out, hipStream_t st) {
    if (B <= 0 || C <= 0) return ctx->fail(SR_ERR_INVALID, "ssim: empty tensor");
    if (H < 11 || W < 11) return ctx->fail(SR_ERR_INVALID, "ssim: H and W must be >= 11 (filter_size)");
    GaussW gw;
    double g[11], s = 0.0;
    for (int i = 0; i < 11; ++i) { const double x = i - 5.0; g[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < 11; ++i) gw.g[i] = (float)(g[i] / s);
    const int oH = H - 10, oW = W - 10;
    const int tilesX = (oW + ST - 1) / ST, tilesY = (oH + ST - 1) / ST;
    const int nblk = tilesX * tilesY;
    float* partial;
    int rc = reduce_scratch(ctx, (size_t)B * nblk, &partial);
    if (rc) return rc;
    const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(nblk, B), dim3(256), 0, st, a, b, H, W, C, gw, c1, c2, tilesX, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(B), dim3(256), 0, st, partial, nblk, (double)oH * oW * C, 1.f, 2, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int extract_patches_launch(sr_ctx* ctx, const float* img, int H, int W, int C, int patch, int stride, float mul, float add, int out_dtype,
                           void* out, int ny, int nx, hipStream_t st) {
    const int64_t n = (int64_t)ny * nx * patch * patch * C;
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(extract_patches_kernel, dim3(grid_for(n)), dim3(256), 0, st, img, H, W, C, patch, stride, mul, add, out_dtype, out,
                       ny, nx);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int overlap_add_launch(sr_ctx* ctx, const void* patches, int in_dtype, int H, int W, int C, int patch, int stride, int scale, float mul,
                       float add, int ny, int nx, float* out, hipStream_t st) {
    const int64_t n = (int64_t)H * scale * W * scale * C;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(grid_for(n)), dim3(256), 0, st, patches, in_dtype, H, W, C, patch, stride, scale, mul, add,
                       ny, nx, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Clock probe (sr_measure_clock): the shader clock the chip holds under a dense bf16 MFMA load, from
// delta(s_memtime) / delta(s_memrealtime) x 100 MHz around an MFMA loop (MI355X_MICROARCH.md, DVFS give-back item 6).
// bench.py prices its roofline peak at this clock beside the nominal one.  Stamps go to a buffer nothing else reads.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) clock_probe_kernel(unsigned long long* stamps, float* sink, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u; a[i] = (bf16_t)((float)((s >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
        s = s * 1664525u + 1013904223u; b[i] = (bf16_t)((float)((s >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f);
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (v == 123.456f) sink[0] = v;                       // keeps the MFMAs alive; never true in practice
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    (void)lane;
}

int clock_probe_launch(sr_ctx* ctx, float* mhz_out, hipStream_t st) {
    const int nblk = 1024, iters = 4000;                  // 1024 workgroups x 4 waves x 128k MFMAs: ~2 ms per launch
    unsigned long long* d = static_cast<unsigned long long*>(ctx->dalloc(sizeof(unsigned long long) * 2 * nblk + sizeof(float)));
    if (!d) return SR_ERR_OOM;
    float* sink = reinterpret_cast<float*>(d + 2 * nblk);
    for (int rep = 0; rep < 150; ++rep)                   // ~0.3 s of back-to-back load before the launch that is read
        hipLaunchKernelGGL(clock_probe_kernel, dim3(nblk), dim3(256), 0, st, d, sink, iters, 12345u + rep);
    std::vector<unsigned long long> h(2 * nblk);
    hipError_t e = hipMemcpyAsync(h.data(), d, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx->dfree(d);
    if (e != hipSuccess) return ctx->fail(SR_ERR_HIP, std::string("clock probe: ") + hipGetErrorString(e));
    std::vector<double> f;
    for (int i = 0; i < nblk; ++i) if (h[2 * i + 1] > 0) f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
    if (f.empty()) return ctx->fail(SR_ERR_HIP, "clock probe: no stamps");
    std::sort(f.begin(), f.end());
    *mhz_out = (float)f[f.size() / 2];
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// cv2.resize for INTER_LINEAR / INTER_AREA / INTER_LANCZOS4 (classic_algorithms.py:7-21, the per-file codes of
// interpolation_map.pkl: loading_methods.py:131-148).  Two tiny kernels build the per-axis tap tables (clamped source index +
// weight, OpenCV's resizeGeneric / interpolateLanczos4 / computeResizeAreaTab formulas), one kernel applies them: horizontal taps
// first, rounded to the row type, then the vertical taps -- the order of OpenCV's two passes.
// ------------------------------------------------------------------------------------------------
constexpr int RS_MAXT = 16;

__global__ void resize_taps_kernel(int n_src, int n_dst, int interp, int area_up, int T, int* idx, float* w, int* iw) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dst) return;
    const double inv_scale = (double)n_dst / (double)n_src, scale = 1.0 / inv_scale;
    int* id = idx + (size_t)d * T;
    float* wd = w + (size_t)d * T;
    for (int k = 0; k < T; ++k) { id[k] = 0; wd[k] = 0.f; }
    if (interp == 3 && !area_up) {                                 // INTER_AREA, shrinking
        // (explicit *_rn operations throughout: a contracted fma changes which side of an integer a coordinate falls on -- OpenCV's
        //  x86 builds round the product first)
        const double f1 = __dmul_rn((double)d, scale), f2 = __dadd_rn(f1, scale);
        const double cell = fmin(scale, (double)n_src - f1);
        int s1 = (int)ceil(f1), s2 = (int)floor(f2);
        s2 = min(s2, n_src - 1);
        s1 = min(s1, s2);
        int k = 0;
        if (s1 - f1 > 1e-3) { id[k] = s1 - 1; wd[k] = (float)((s1 - f1) / cell); ++k; }
        for (int sx = s1; sx < s2 && k < T; ++sx, ++k) { id[k] = sx; wd[k] = (float)(1.0 / cell); }
        if (f2 - s2 > 1e-3 && k < T) { id[k] = s2; wd[k] = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell); ++k; }
    } else if (interp == 1 || interp == 3) {                       // linear taps; INTER_AREA enlarging uses "area" coordinates
        int s; float f;
        if (interp == 3) {
            s = (int)floor(__dmul_rn((double)d, scale));
            f = (float)__dsub_rn((double)(d + 1), __dmul_rn((double)(s + 1), inv_scale));
            f = f <= 0.f ? 0.f : f - floorf(f);
        } else {
            const float fx = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
            s = (int)floorf(fx);
            f = fx - (float)s;
        }
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        id[0] = min(max(s, 0), n_src - 1); id[1] = min(max(s + 1, 0), n_src - 1);
        wd[0] = 1.f - f; wd[1] = f;
    } else {                                                       // INTER_LANCZOS4: 8 taps at s-3 .. s+4
        const float fx = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
        int s = (int)floorf(fx);
        float x = fx - (float)s;
        if (x >= 1.f) { s += 1; x = 0.f; }                           // fx a hair below an integer: (s, 1.0) is (s + 1, 0.0) -- and 1.0 would divide by zero below
        for (int k = 0; k < 8; ++k) id[k] = min(max(s - 3 + k, 0), n_src - 1);
        if (x < 1.1920929e-07f) { wd[3] = 1.f; }
        else {
            const double s45 = 0.70710678118654752440084436210485;
            const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
            const double y0 = -((double)x + 3) * 3.14159265358979323846 * 0.25, s0 = sin(y0), c0 = cos(y0);
            float sum = 0.f, c[8];
            for (int k = 0; k < 8; ++k) {
                const double y = -((double)x + 3 - k) * 3.14159265358979323846 * 0.25;
                c[k] = (float)((cs[k][0] * s0 + cs[k][1] * c0) / (y * y));
                sum += c[k];
            }
            sum = 1.f / sum;
            for (int k = 0; k < 8; ++k) wd[k] = c[k] * sum;
        }
    }
    if (iw) for (int k = 0; k < T; ++k) {                          // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)
        const float v = rintf(wd[k] * 2048.f);
        iw[(size_t)d * T + k] = (int)fminf(fmaxf(v, -32768.f), 32767.f);
    }
}

__global__ void resize_apply_f32_kernel(const float* x, int B, int H, int W, int C, int oH, int oW, int TX, int TY, const int* ix, const float* wx,
                                        const int* iy, const float* wy, float* y) {
    const int64_t n = (int64_t)B * oH * oW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int b = (int)(t / oH);
        float acc = 0.f;
        for (int j = 0; j < TY; ++j) {
            const float wyj = wy[oy * TY + j];
            const float* row = x + ((int64_t)b * H + iy[oy * TY + j]) * W * C + c;
            float r = 0.f;
            for (int k = 0; k < TX; ++k) r = __fadd_rn(r, __fmul_rn(row[(int64_t)ix[ox * TX + k] * C], wx[ox * TX + k]));
            acc = __fadd_rn(acc, __fmul_rn(r, wyj));
        }
        y[i] = acc;
    }
}

__global__ void resize_apply_u8_kernel(const uint8_t* x, int B, int H, int W, int C, int oH, int oW, int TX, int TY, const int* ix, const int* iwx,
                                       const int* iy, const int* iwy, uint8_t* y) {
    const int64_t n = (int64_t)B * oH * oW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int b = (int)(t / oH);
        long long rows[RS_MAXT];
        for (int j = 0; j < TY; ++j) {
            const uint8_t* row = x + ((int64_t)b * H + iy[oy * TY + j]) * W * C + c;
            long long r = 0;
            for (int k = 0; k < TX; ++k) r += (long long)row[(int64_t)ix[ox * TX + k] * C] * iwx[ox * TX + k];
            rows[j] = r;
        }
        long long v;
        if (TX == 2 && TY == 2) v = (((iwy[oy * 2] * (rows[0] >> 4)) >> 16) + ((iwy[oy * 2 + 1] * (rows[1] >> 4)) >> 16) + 2) >> 2;   // VResizeLinear<uchar>
        else {
            v = 0;
            for (int j = 0; j < TY; ++j) v += rows[j] * iwy[oy * TY + j];
            v = (v + (1 << 21)) >> 22;
        }
        y[i] = (uint8_t)min(max(v, 0ll), 255ll);
    }
}

// INTER_NEAREST (OpenCV resizeNN): source index = min(cvFloor(d * (1 / (n_dst / n_src))), n_src - 1) in double; a gather of ES-byte elements
template <typename E>
__global__ void resize_nearest_kernel(const E* x, int B, int H, int W, int C, int oH, int oW, double ify, double ifx, E* y) {
    const int64_t n = (int64_t)B * oH * oW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int b = (int)(t / oH);
        const int sx = min((int)floor(__dmul_rn((double)ox, ifx)), W - 1), sy = min((int)floor(__dmul_rn((double)oy, ify)), H - 1);
        y[i] = x[(((int64_t)b * H + sy) * W + sx) * C + c];
    }
}

// uint8 INTER_AREA by whole-number factors on both axes (OpenCV resizeAreaFast_<uchar, int, ...>): integer sum over the fy x fx cell, then
// (sum + 2) >> 2 for the 2 x 2 cell of 1-, 3- and 4-channel images (ResizeAreaFastVec), saturate_cast<uchar>(sum * (1.f / area)) otherwise
// (a float product, rounded half to even).
__global__ void resize_area_fast_u8_kernel(const uint8_t* x, int B, int H, int W, int C, int oH, int oW, int fy, int fx, uint8_t* y) {
    const int64_t n = (int64_t)B * oH * oW * C;
    const bool shift2 = fy == 2 && fx == 2 && (C == 1 || C == 3 || C == 4);
    const float scale = 1.f / (float)(fy * fx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int b = (int)(t / oH);
        int sum = 0;
        for (int j = 0; j < fy; ++j) {
            const uint8_t* row = x + (((int64_t)b * H + oy * fy + j) * W + (int64_t)ox * fx) * C + c;
            for (int k = 0; k < fx; ++k) sum += row[(int64_t)k * C];
        }
        y[i] = shift2 ? (uint8_t)((sum + 2) >> 2) : (uint8_t)fminf(fmaxf(rintf(__fmul_rn((float)sum, scale)), 0.f), 255.f);
    }
}

// uint8 INTER_AREA by any other shrink factor (OpenCV resizeArea_<uchar, float>): the float path's taps and order of operations on the uint8
// values, the float sum rounded half to even and saturated.
__global__ void resize_apply_area_u8_kernel(const uint8_t* x, int B, int H, int W, int C, int oH, int oW, int TX, int TY, const int* ix, const float* wx,
                                            const int* iy, const float* wy, uint8_t* y) {
    const int64_t n = (int64_t)B * oH * oW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int b = (int)(t / oH);
        float acc = 0.f;
        for (int j = 0; j < TY; ++j) {
            const float wyj = wy[oy * TY + j];
            const uint8_t* row = x + ((int64_t)b * H + iy[oy * TY + j]) * W * C + c;
            float r = 0.f;
            for (int k = 0; k < TX; ++k) r = __fadd_rn(r, __fmul_rn((float)row[(int64_t)ix[ox * TX + k] * C], wx[ox * TX + k]));
            acc = __fadd_rn(acc, __fmul_rn(r, wyj));
        }
        y[i] = (uint8_t)fminf(fmaxf(rintf(acc), 0.f), 255.f);
    }
}

int resize_launch(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW, int interp, void* y, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || outH <= 0 || outW <= 0) return ctx->fail(SR_ERR_INVALID, "resize: empty tensor");
    if (dtype != SR_DTYPE_F32 && dtype != SR_DTYPE_U8) return ctx->fail(SR_ERR_INVALID, "resize: dtype must be f32 or u8");
    if (interp == 2) return bicubic_launch(ctx, x, dtype, B, H, W, C, outH, outW, y, dtype, C, st);
    const int64_t n_out = (int64_t)B * outH * outW * C;
    if (interp == 0) {
        const double ify = 1.0 / ((double)outH / (double)H), ifx = 1.0 / ((double)outW / (double)W);
        if (dtype == SR_DTYPE_F32)
            hipLaunchKernelGGL(resize_nearest_kernel<float>, dim3(grid_for(n_out)), dim3(256), 0, st, static_cast<const float*>(x), B, H, W, C, outH, outW, ify, ifx, static_cast<float*>(y));
        else
            hipLaunchKernelGGL(resize_nearest_kernel<uint8_t>, dim3(grid_for(n_out)), dim3(256), 0, st, static_cast<const uint8_t*>(x), B, H, W, C, outH, outW, ify, ifx, static_cast<uint8_t*>(y));
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    if (interp != 1 && interp != 3 && interp != 4)
        return ctx->fail(SR_ERR_INVALID, "resize: interpolation must be INTER_NEAREST (0), INTER_LINEAR (1), INTER_CUBIC (2), INTER_AREA (3) or INTER_LANCZOS4 (4)");
    if (interp == 1 && W == 2 * outW && H == 2 * outH) interp = 3;          // OpenCV's resize(): bilinear halving IS the 2 x 2 box mean
    const bool shrink_both = outW <= W && outH <= H;
    const int area_up = interp == 3 && !shrink_both;
    if (dtype == SR_DTYPE_U8 && interp == 3 && !area_up && W % outW == 0 && H % outH == 0) {
        hipLaunchKernelGGL(resize_area_fast_u8_kernel, dim3(grid_for(n_out)), dim3(256), 0, st, static_cast<const uint8_t*>(x), B, H, W, C, outH, outW, H / outH, W / outW,
                           static_cast<uint8_t*>(y));
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    auto taps = [&](int ns, int nd) { return interp == 4 ? 8 : (interp == 3 && !area_up) ? (int)ceil((double)ns / nd) + 2 : 2; };
    const int TX = taps(W, outW), TY = taps(H, outH);
    if (TX > RS_MAXT || TY > RS_MAXT) return ctx->fail(SR_ERR_INVALID, "resize: INTER_AREA shrink factor above 14 is not supported");
    const size_t nx = (size_t)outW * TX, ny = (size_t)outH * TY, need = (nx + ny) * 12;
    if (need > ctx->tab_cap) {
        if (ctx->tab_buf) { SR_HIP(ctx, hipDeviceSynchronize()); ctx->dfree(ctx->tab_buf); ctx->tab_buf = nullptr; ctx->tab_cap = 0; }
        ctx->tab_buf = ctx->dalloc(need);
        if (!ctx->tab_buf) return SR_ERR_OOM;
        ctx->tab_cap = need;
    }
    int* ix = static_cast<int*>(ctx->tab_buf);
    float* wx = reinterpret_cast<float*>(ix + nx);
    int* iwx = reinterpret_cast<int*>(wx + nx);
    int* iy = iwx + nx;
    float* wy = reinterpret_cast<float*>(iy + ny);
    int* iwy = reinterpret_cast<int*>(wy + ny);
    hipLaunchKernelGGL(resize_taps_kernel, dim3((outW + 255) / 256), dim3(256), 0, st, W, outW, interp, area_up, TX, ix, wx, iwx);
    hipLaunchKernelGGL(resize_taps_kernel, dim3((outH + 255) / 256), dim3(256), 0, st, H, outH, interp, area_up, TY, iy, wy, iwy);
    const int64_t n = (int64_t)B * outH * outW * C;
    if (dtype == SR_DTYPE_F32)
        hipLaunchKernelGGL(resize_apply_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const float*>(x), B, H, W, C, outH, outW, TX, TY, ix, wx, iy, wy,
                           static_cast<float*>(y));
    else if (interp == 3 && !area_up)
        hipLaunchKernelGGL(resize_apply_area_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const uint8_t*>(x), B, H, W, C, outH, outW, TX, TY, ix, wx, iy, wy,
                           static_cast<uint8_t*>(y));
    else
        hipLaunchKernelGGL(resize_apply_u8_kernel, dim3(grid_for(n)), dim3(256), 0, st, static_cast<const uint8_t*>(x), B, H, W, C, outH, outW, TX, TY, ix, iwx, iy, iwy,
                           static_cast<uint8_t*>(y));
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Pieces of the ESRGAN generator loss (ESRGAN_model.py:401-473) and of its discriminator / VGG19 graphs
// ------------------------------------------------------------------------------------------------
// Keras Conv2D(strides=2, padding="same") == the stride-1 SAME conv sampled at every second position: TF pads
// total = max((ceil(n/2)-1)*2 + 3 - n, 0) with the smaller half in FRONT, so for even n output i sits on input 2i+1 (pad 0 / 1) and
// for odd n on input 2i (pad 1 / 1).  (the discriminator's maps 48 -> 24 -> 12 -> 6.)
__global__ void subsample2_kernel(const void* x, int dtype, int B, int H, int W, int C, int64_t x_cs, void* y, int64_t y_cs, int oH, int oW) {
    const int64_t n = (int64_t)B * oH * oW * C;
    const int offy = (H & 1) ? 0 : 1, offx = (W & 1) ? 0 : 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % oW); t /= oW;
        const int oy = (int)(t % oH);
        const int64_t b = t / oH;
        const int64_t src = ((b * H + 2 * oy + offy) * W + 2 * ox + offx) * x_cs + c, dst = ((b * oH + oy) * oW + ox) * y_cs + c;
        if (dtype == SR_DTYPE_BF16) static_cast<bf16_t*>(y)[dst] = static_cast<const bf16_t*>(x)[src];
        else static_cast<float*>(y)[dst] = static_cast<const float*>(x)[src];
    }
}

int subsample2_launch(sr_ctx* ctx, int dtype, const void* x, int B, int H, int W, int C, int64_t x_cs, void* y, int64_t y_cs, hipStream_t st) {
    const int oH = (H + 1) / 2, oW = (W + 1) / 2;
    const int64_t n = (int64_t)B * oH * oW * C;
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "stride-2 conv: empty tensor");
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, dtype, B, H, W, C, x_cs, y, y_cs, oH, oW);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// _preprocess_vgg_input (ESRGAN_model.py:401-408): [-1,1] RGB -> (x+1)*127.5 -> BGR - ImageNet means (caffe mode)
__global__ void vgg_preproc_kernel(const void* x, int in_dtype, int64_t npix, void* y, int out_dtype, int Cp) {
    const int64_t n = npix * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const int64_t p = i / Cp;
        float v = 0.f;
        if (c < 3) {
            const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
            v = (ld_dt(x, p * 3 + (2 - c), in_dtype) + 1.f) * 127.5f - mean;
        }
        st_dt(y, i, v, out_dtype);
    }
}

int vgg_preproc_launch(sr_ctx* ctx, const void* x, int in_dtype, int64_t npix, void* y, int out_dtype, int Cp, hipStream_t st) {
    if (npix <= 0) return SR_OK;
    hipLaunchKernelGGL(vgg_preproc_kernel, dim3(grid_for(npix * Cp)), dim3(256), 0, st, x, in_dtype, npix, y, out_dtype, Cp);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

__global__ void absdiff_partial_kernel(const float* a, const float* b, int64_t n, float* partial) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += fabsf(a[i] - b[i]);
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

int l1_launch(sr_ctx* ctx, const float* a, const float* b, int64_t n, float* out, hipStream_t st) {
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "l1: empty tensor");
    const int nblk = (int)grid_for(n) > 1024 ? 1024 : (int)grid_for(n);
    float* partial;
    int rc = reduce_scratch(ctx, (size_t)nblk, &partial);
    if (rc) return rc;
    hipLaunchKernelGGL(absdiff_partial_kernel, dim3(nblk), dim3(256), 0, st, a, b, n, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, partial, nblk, (double)n, 1.f, 1, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// _spectral_loss (ESRGAN_model.py:461-473): tf.signal.fft2d works on the innermost two axes -- (W, C) of an NHWC batch.
// One workgroup per image row (b, h): the 3-point DFT over the channels per pixel, then for each of the W x 3 output bins the W-point
// DFT by direct summation against a twiddle table (W is 48..192 here: a few hundred k MACs per row); sum | |F(a)| - |F(b)| |.
__global__ void __launch_bounds__(256) spectral_wc_partial_kernel(const float* a, const float* b, int W, float* partial) {
    extern __shared__ float sm[];
    float* tw = sm;                       // [W][2]   cos, -sin of 2 pi k / W
    float* ga = tw + 2 * W;               // [W][3][2] channel DFT of image a
    float* gb = ga + 6 * W;
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const float* ra = a + row * W * 3;
    const float* rb = b + row * W * 3;
    for (int k = threadIdx.x; k < W; k += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)W, &s, &c);
        tw[2 * k] = (float)c; tw[2 * k + 1] = (float)(-s);
        const float c3 = -0.5f, s3 = 0.86602540378443864676f;          // exp(-2 pi i / 3) = c3 - i s3
        for (int im = 0; im < 2; ++im) {
            const float* r = im ? rb : ra;
            float* g = im ? gb : ga;
            const float x0 = r[3 * k], x1 = r[3 * k + 1], x2 = r[3 * k + 2];
            g[6 * k + 0] = x0 + x1 + x2;                  g[6 * k + 1] = 0.f;
            g[6 * k + 2] = x0 + c3 * (x1 + x2);           g[6 * k + 3] = -s3 * (x1 - x2);     // v = 1: x1 e^{-2pi i/3} + x2 e^{-4pi i/3}
            g[6 * k + 4] = x0 + c3 * (x1 + x2);           g[6 * k + 5] = s3 * (x1 - x2);      // v = 2: the conjugate for real input
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int o = threadIdx.x; o < 3 * W; o += blockDim.x) {
        const int u = o / 3, v = o - 3 * u;
        float are = 0.f, aim = 0.f, bre = 0.f, bim = 0.f;
        int k = 0;                                     // (u * w) mod W, incrementally
        for (int w = 0; w < W; ++w) {
            const float tc = tw[2 * k], ts = tw[2 * k + 1];
            const float gar = ga[6 * w + 2 * v], gai = ga[6 * w + 2 * v + 1], gbr = gb[6 * w + 2 * v], gbi = gb[6 * w + 2 * v + 1];
            are += gar * tc - gai * ts; aim += gar * ts + gai * tc;
            bre += gbr * tc - gbi * ts; bim += gbr * ts + gbi * tc;
            k += u; if (k >= W) k -= W;
        }
        acc += fabsf(sqrtf(are * are + aim * aim) - sqrtf(bre * bre + bim * bim));
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[row] = acc;
}

int spectral_l1_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float* out, hipStream_t st) {
    if (C != 3) return ctx->fail(SR_ERR_INVALID, "spectral loss: built for 3 channels (the FFT runs over the (W, C) axes)");
    // the kernel keeps one image row and its twiddles in LDS: 56 bytes per pixel of width, 160 KiB per CU
    constexpr int SPECTRAL_MAX_W = 160 * 1024 / 56;
    if (B <= 0 || H <= 0 || W <= 0) return ctx->fail(SR_ERR_INVALID, "spectral loss: bad shape");
    if (W > SPECTRAL_MAX_W) return ctx->fail(SR_ERR_INVALID, "spectral loss: image width " + std::to_string(W) + " exceeds " + std::to_string(SPECTRAL_MAX_W) + " (one row + twiddles must fit a CU's 160 KiB of LDS)");
    const int64_t rows = (int64_t)B * H;
    float* partial;
    int rc = reduce_scratch(ctx, (size_t)rows, &partial);
    if (rc) return rc;
    const size_t lds = sizeof(float) * (size_t)(2 * W + 12 * W);
    auto kern = spectral_wc_partial_kernel;
    if (lds > 48 * 1024) { if (int r2 = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)) return r2; }
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), lds, st, a, b, W, partial);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, st, partial, (int)rows, (double)rows * W * 3, 1.f, 1, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}
