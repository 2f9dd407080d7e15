// train_ops.hip -- backward-pass pieces for SRCNN.fit / EDSR.fit (SRCNN_model.py:55-98, EDSR_model.py:127-176; Keras model.fit with
// loss = mean_squared_error) and, later, the ESRGAN training step (ESRGAN_model.py:475-533).  fp32 throughout: the reference trains in
// fp32.
//   * dgrad needs no kernel of its own: for a stride-1 SAME conv, dX = conv(dY, W rotated by 180 degrees with the channel axes
//     swapped) -- the host hands the forward kernels the transformed weights (sr355/train.py).
//   * wgrad: dW[ky,kx,ci,co] = sum over pixels of X[b, y+ky-p, x+kx-p, ci] * dY[b,y,x,co] -- per tap a [Cin x P] x [P x Cout] product
//     with the pixels as the reduction axis, on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain): a lane feeds one channel of one of two
//     consecutive pixels straight from global memory (32 consecutive floats per half-wave: whole 128-byte lines, no LDS).  The pixel
//     range is split over workgroups; partial 32x32 tiles are summed in a second pass in a fixed order (reproducible, no atomics).
//   * element-wise backward ops and the space_to_depth that undoes depth_to_space (TF "DCR" order).
#include "common.h"

namespace {

constexpr int WG_SPLIT_PIX = 2048;        // pixels per workgroup slice of the reduction axis

__global__ void __launch_bounds__(256) wgrad_partial_kernel(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int KS,
                                                            int nci, int nco, int nsplit, float* partial) {
    // blockIdx.x = ((tap * nci) + cib) * nco + cob, blockIdx.y = split
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, k = lane >> 5;
    int t = blockIdx.x;
    const int cob = t % nco; t /= nco;
    const int cib = t % nci;
    const int tap = t / nci;
    const int ky = tap / KS, kx = tap - ky * KS, pad = (KS - 1) / 2;
    const int ci = cib * 32 + i, co = cob * 32 + i;
    const bool ci_ok = ci < Cin, co_ok = co < Cout;
    const int64_t P = (int64_t)B * H * W;
    const int64_t per = (P + nsplit - 1) / nsplit;
    const int64_t p0 = (int64_t)blockIdx.y * per, p1 = min(P, p0 + per);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // The 4 waves interleave pixel pairs; a lane walks pixels pp = p0 + 2 wave + k, + 8, + 16, ...  Round 3: the pixel's (row, column) is
    // carried along instead of being divided out of pp every time, and eight pixel pairs' operands are requested before the eight MFMAs that
    // consume them (round 2: one dependent pair of loads in front of every MFMA -- a launch ran at the latency of 256 serial loads, 151 us
    // for 0.2 GFLOP).  The MFMAs still run in pixel order, so the sums are bit for bit the ones of round 2.
    constexpr int U = 8;
    int64_t pp = p0 + 2 * wave + k;
    int xq = (int)(pp % W);
    int64_t r = pp / W;                     // image row counted through the batch
    int yq = (int)(r % H);
    for (int64_t pb = p0 + 2 * wave; pb < p1; pb += 8 * U) {
        float a[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = pp < p1;
            const int sy = yq + ky - pad, sx = xq + kx - pad;
            const bool in = live && ci_ok && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            const float av = x[in ? ((r - yq + sy) * W + sx) * Cin + ci : 0];
            const float bw = dy[(live && co_ok) ? pp * Cout + co : 0];
            a[u] = in ? av : 0.f;
            bv[u] = (live && co_ok) ? bw : 0.f;
            pp += 8;
            xq += 8;
            while (xq >= W) { xq -= W; ++r; if (++yq == H) yq = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u], acc, 0, 0, 0);
    }
    // sum the 4 waves' tiles through LDS, fixed order
    __shared__ float red[4][32 * 32];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * k;      // C/D layout of the 32x32 MFMA: col = lane & 31, row from the register index
        red[wave][row * 32 + i] = acc[e];
    }
    __syncthreads();
    float* out = partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 1024);
    for (int e = threadIdx.x; e < 1024; e += 256) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// 3x3 layers, LDS-tiled (round 4).  The per-tap kernel above feeds every MFMA from two 4-byte global loads per lane and runs at their latency (56 us for the
// 2 GFLOP of a dense block's conv5: 23 % of the fp32 peak; three variants of it -- two batches in flight, more pixel slices, nine taps per workgroup with
// address arithmetic per load -- measured slower).  Here a workgroup owns an (input block, cout block) pair and a run of 8 x 24-pixel image tiles: the x tile with
// its halo (10 x 26 pixels x 32 channels) and the dy tile (192 pixels x 32 couts) are staged into LDS with 16-byte loads, and every dy value meets the NINE shifted x
// values of its pixel in nine accumulator tiles -- ten conflict-free ds_read_b32 per nine MFMAs.  The four waves interleave the tile's pixel pairs; their nine
// tiles are summed through LDS in wave order, the runs of tiles (pixel splits) in wgrad_finish_kernel in split order: deterministic, no atomics.
constexpr int WT_H = 8, WT_W = 24, WT_PW = WT_W + 2, WT_PH = WT_H + 2, WT_NPIX = WT_H * WT_W;

__global__ void __launch_bounds__(256) wgrad3_tile_kernel(const float* x, int64_t x_cs, const float* dy, int64_t dy_cs, int B, int H, int W, int Cin, int Cout, int nci, int nco,
                                                          int tiles_per_wg, int tilesX, int tilesY, float* partial, double* bpart) {
    // x / dy: NHWC views -- x_cs / dy_cs channels per pixel in the underlying buffer, the pointers already at the view's first channel (round 4: the
    // trainer's dense blocks keep their concat tensor in ONE buffer and hand every conv a channel range of it)
    __shared__ __attribute__((aligned(16))) float xt[WT_PH * WT_PW * 32];
    __shared__ __attribute__((aligned(16))) float dt[WT_NPIX * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, k = lane >> 5;
    const int cob = blockIdx.x % nco, cib = blockIdx.x / nco;
    const int ci0 = cib * 32, co0 = cob * 32;
    const int ntile = B * tilesY * tilesX;
    const int t0 = blockIdx.y * tiles_per_wg, t1 = min(ntile, t0 + tiles_per_wg);
    const bool vec_x = (Cin & 3) == 0 && (x_cs & 3) == 0 && ((uintptr_t)x & 15) == 0, vec_y = (Cout & 3) == 0 && (dy_cs & 3) == 0 && ((uintptr_t)dy & 15) == 0;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const bool do_b = bpart != nullptr && cib == 0;
    double bsum = 0.0;                                             // threads < 256: cout tid & 31, pixel group tid >> 5
    for (int tile = t0; tile < t1; ++tile) {
        int tt = tile;
        const int tx = tt % tilesX; tt /= tilesX;
        const int ty = tt % tilesY;
        const int b = tt / tilesY;
        const int y0 = ty * WT_H, x0 = tx * WT_W;
        __syncthreads();                                           // the previous tile's operands have been read
        for (int u = tid; u < WT_PH * WT_PW * 8; u += 256) {
            const int pix = u >> 3, sl = u & 7;
            const int py = pix / WT_PW, px = pix - py * WT_PW;
            const int gy = y0 + py - 1, gx = x0 + px - 1, c = ci0 + sl * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin) {
                const float* src = x + (((int64_t)b * H + gy) * W + gx) * x_cs + c;
                if (vec_x) v = *reinterpret_cast<const f32x4*>(src);
                else
                    for (int e = 0; e < 4; ++e) v[e] = c + e < Cin ? src[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(xt + pix * 32 + sl * 4) = v;
        }
        for (int u = tid; u < WT_NPIX * 8; u += 256) {
            const int pix = u >> 3, sl = u & 7;
            const int py = pix / WT_W, px = pix - py * WT_W;
            const int gy = y0 + py, gx = x0 + px, c = co0 + sl * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W && c < Cout) {
                const float* src = dy + (((int64_t)b * H + gy) * W + gx) * dy_cs + c;
                if (vec_y) v = *reinterpret_cast<const f32x4*>(src);
                else
                    for (int e = 0; e < 4; ++e) v[e] = c + e < Cout ? src[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dt + pix * 32 + sl * 4) = v;
        }
        __syncthreads();
        if (do_b) {
            const int co = tid & 31;
            for (int p = tid >> 5; p < WT_NPIX; p += 8) bsum += (double)dt[p * 32 + co];
        }
        for (int kk = wave; kk < WT_NPIX / 2; kk += 4) {
            const int p = 2 * kk + k;                              // this lane's pixel of the pair (24 is even: both in one row)
            const int py = p / WT_W, px = p - py * WT_W;
            const float bv = dt[p * 32 + i];
            const float* xb = xt + (py * WT_PW + px) * 32 + i;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[(ky * WT_PW + kx) * 32], bv, acc[ky * 3 + kx], 0, 0, 0);
        }
    }
    float* red = xt;                                               // 4 x 1024 floats: the x tile's space (16 of its 33 KiB)
    static_assert(WT_PH * WT_PW * 32 >= 4 * 1024, "reduction fits the x tile");
    const int ntiles_out = 9 * nci * nco;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * k;        // C/D layout of the 32x32 MFMA: col = lane & 31, row from the register index
            red[wave * 1024 + row * 32 + i] = acc[t][e];
        }
        __syncthreads();
        float* out = partial + (((size_t)blockIdx.y * ntiles_out + (t * nci + cib) * nco + cob) * 1024);
        for (int e = tid; e < 1024; e += 256) out[e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    }
    if (do_b) {
        __syncthreads();
        double* bred = reinterpret_cast<double*>(dt);              // [8][32] doubles
        bred[(tid >> 5) * 32 + (tid & 31)] = bsum;
        __syncthreads();
        if (tid < 32) {
            double t2 = 0.0;
            for (int j = 0; j < 8; ++j) t2 += bred[j * 32 + tid];
            bpart[((size_t)blockIdx.y * nco + cob) * 32 + tid] = t2;
        }
    }
}

__global__ void wgrad_finish_kernel(const float* partial, int ntiles, int nsplit, int Cin, int Cout, int nci, int nco, float* dw, const double* bpart, float* db) {
    // one thread per element of a quarter (blockIdx.y) of one 32x32 tile; dw is HWIO [tap][Cin][Cout].  (Round 4: four elements per thread on a grid of `ntiles`
    // workgroups -- 45 for a 160 -> 32 conv -- were four dependent rounds of loads on a sixth of the chip: 9 us per conv, 3.3 ms of the cfg3 step.)
    const int tile = blockIdx.x;
    int t = tile;
    const int cob = t % nco; t /= nco;
    const int cib = t % nci;
    const int tap = t / nci;
    {
        const int e = blockIdx.y * 256 + threadIdx.x;
        const int ci = cib * 32 + e / 32, co = cob * 32 + (e & 31);
        if (ci < Cin && co < Cout) {
            float s = 0.f;
#pragma unroll 16
            for (int sp = 0; sp < nsplit; ++sp) s += partial[((size_t)sp * ntiles + tile) * 1024 + e];      // (unrolled: the loads of 16 splits in flight at once)
            dw[((size_t)tap * Cin + ci) * Cout + co] = s;
        }
    }
    if (bpart && db && tap == 0 && cib == 0 && blockIdx.y == 0 && threadIdx.x < 32 && cob * 32 + (int)threadIdx.x < Cout) {      // the tiled kernel's bias-gradient partials
        double t2 = 0.0;
#pragma unroll 16
        for (int sp = 0; sp < nsplit; ++sp) t2 += bpart[((size_t)sp * nco + cob) * 32 + threadIdx.x];
        db[cob * 32 + threadIdx.x] = (float)t2;
    }
}

__global__ void colsum_kernel(const float* dy, int64_t P, int C, float* out) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int64_t p = threadIdx.x; p < P; p += blockDim.x) s += (double)dy[p * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[c] = (float)red[0];
}

__device__ __forceinline__ float eltwise_value(int op, float av, float bv, float alpha, float beta) {
    switch (op) {
        case SR_ELT_AXPBY: return alpha * av + beta * bv;
        case SR_ELT_RELU_BWD: return bv > 0.f ? av : 0.f;
        case SR_ELT_LRELU_BWD: return bv > 0.f ? av : 0.2f * av;
        case SR_ELT_CLIP01_BWD: return (bv >= 0.f && bv <= 1.f) ? av : 0.f;
        case SR_ELT_MUL: return alpha * av * bv;
        case SR_ELT_TANH_BWD: return av * (1.f - bv * bv);
        case SR_ELT_CLIP01: return fminf(fmaxf(av, 0.f), 1.f);
        case SR_ELT_SIGN_DIFF: return alpha * (av > bv ? 1.f : (av < bv ? -1.f : 0.f));
        default: return 0.f;
    }
}

// the same ops on channel ranges of NHWC buffers: element (pixel p, channel c) of a view lives at p * cs + c from the view's pointer
__global__ void eltwise_view_kernel(int op, const float* a, int64_t a_cs, const float* b, int64_t b_cs, float alpha, float beta, float* out, int64_t o_cs, int64_t npix, int C) {
    const int64_t n = npix * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C;
        const int c = (int)(i - p * C);
        out[p * o_cs + c] = eltwise_value(op, a[p * a_cs + c], b ? b[p * b_cs + c] : 0.f, alpha, beta);
    }
}

__global__ void eltwise_kernel(int op, const float* a, const float* b, float alpha, float beta, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float av = a[i], bv = b ? b[i] : 0.f;
        float v;
        switch (op) {
            case SR_ELT_AXPBY: v = alpha * av + beta * bv; break;
            case SR_ELT_RELU_BWD: v = bv > 0.f ? av : 0.f; break;                       // a = dy, b = y (post-activation)
            case SR_ELT_LRELU_BWD: v = bv > 0.f ? av : 0.2f * av; break;
            case SR_ELT_CLIP01_BWD: v = (bv >= 0.f && bv <= 1.f) ? av : 0.f; break;    // tf.clip_by_value: the gradient passes inside [min, max]; b = pre-clip value
            case SR_ELT_MUL: v = alpha * av * bv; break;
            case SR_ELT_TANH_BWD: v = av * (1.f - bv * bv); break;                     // b = tanh output
            case SR_ELT_CLIP01: v = fminf(fmaxf(av, 0.f), 1.f); break;
            case SR_ELT_SIGN_DIFF: v = alpha * (av > bv ? 1.f : (av < bv ? -1.f : 0.f)); break;   // d/da mean|a - b| up to the 1/n in alpha
            default: v = 0.f;
        }
        out[i] = v;
    }
}

// inverse of tf.nn.depth_to_space (DCR): x [B, H*r, W*r, C] -> y [B, H, W, (i*r + j)*C + c]
__global__ void space_to_depth_kernel(const float* x, int B, int H, int W, int C, int r, float* y) {
    const int64_t n = (int64_t)B * H * W * r * r * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cc = (int)(idx % (r * r * C));
        int64_t t = idx / (r * r * C);
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int64_t b = t / H;
        const int sub = cc / C, c = cc - sub * C, i = sub / r, j = sub - i * r;
        y[idx] = x[(((b * H * r) + (int64_t)h * r + i) * ((int64_t)W * r) + (int64_t)w * r + j) * C + c];
    }
}

// C[b] = alpha * op(A[b]) op(B[b]), fp32 row-major, op = identity or transpose: the materialised attention of the TRAINING patches (N = 576 / 2304 tokens,
// ESRGAN_model.py:57-65) and its backward products.  64 x 64 tile per workgroup, 16 k per stage through LDS; round 4: the products run on v_mfma_f32_32x32x2_f32
// (a wave owns a 32 x 32 quarter of the tile: 8 MFMAs per stage where 256 lanes x 16 v_fma stood: 270 -> ~70 us for the 2304-token products) and a
// non-transposed operand is staged with the lanes running along k, its contiguous axis (round 3 read it with a stride of K floats between lanes).
// Not a hot-path kernel: inference attention is the streaming kernel of attention.hip.
__global__ void __launch_bounds__(256) sgemm_kernel(const float* A, const float* Bm, float* C, int M, int N, int K, int tA, int tB, float alpha,
                                                    int64_t sA, int64_t sB, int64_t sC) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const float* a = A + (int64_t)blockIdx.z * sA;
    const float* b = Bm + (int64_t)blockIdx.z * sB;
    float* c = C + (int64_t)blockIdx.z * sC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, k = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // a thread stages elements e = tid + 256 j (j < 4) of either tile; the next stage's values are requested before this stage's MFMAs (the kernel
    // ran at one global round trip per 16 k: 240 us for the 2304-token products)
    float pa[4], pb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = threadIdx.x + 256 * j;
            {   // A tile: As[kk][mm] = op(A)[m0 + mm][k0 + kk]
                const int kk = tA ? e >> 6 : e & 15, mm = tA ? e & 63 : e >> 4;
                const int gm = m0 + mm, gk = k0 + kk;
                pa[j] = (gm < M && gk < K) ? (tA ? a[(int64_t)gk * M + gm] : a[(int64_t)gm * K + gk]) : 0.f;
            }
            {   // B tile: Bs[kk][nn] = op(B)[k0 + kk][n0 + nn]
                const int kk = tB ? e & 15 : e >> 6, nn = tB ? e >> 4 : e & 63;
                const int gn = n0 + nn, gk = k0 + kk;
                pb[j] = (gn < N && gk < K) ? (tB ? b[(int64_t)gn * K + gk] : b[(int64_t)gk * N + gn]) : 0.f;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = threadIdx.x + 256 * j;
            As[tA ? e >> 6 : e & 15][tA ? e & 63 : e >> 4] = pa[j];
            Bs[tB ? e & 15 : e >> 6][tB ? e >> 4 : e & 63] = pb[j];
        }
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + k][wm + i], Bs[kk + k][wn + i], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int gm = m0 + wm + (e & 3) + 8 * (e >> 2) + 4 * k, gn = n0 + wn + i;      // C/D layout of the 32x32 MFMA: column = lane & 31, row from the register index
        if (gm < M && gn < N) c[(int64_t)gm * N + gn] = alpha * acc[e];
    }
}

// row softmax (in place) and its backward ds = p * (dp - sum_j dp_j p_j): one workgroup per row
__global__ void softmax_rows_kernel(float* s, int cols) {
    __shared__ float red[256];
    float* r = s + (int64_t)blockIdx.x * cols;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) mx = fmaxf(mx, r[j]);
    red[threadIdx.x] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) { const float e = expf(r[j] - mx); r[j] = e; sum += e; }
    red[threadIdx.x] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float inv = 1.f / red[0];
    for (int j = threadIdx.x; j < cols; j += blockDim.x) r[j] *= inv;
}

__global__ void softmax_bwd_rows_kernel(const float* p, const float* dp, float* ds, int cols) {
    __shared__ float red[256];
    const float* pr = p + (int64_t)blockIdx.x * cols;
    const float* dr = dp + (int64_t)blockIdx.x * cols;
    float* o = ds + (int64_t)blockIdx.x * cols;
    float sum = 0.f;
    for (int j = threadIdx.x; j < cols; j += blockDim.x) sum += pr[j] * dr[j];
    red[threadIdx.x] = sum; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    const float dot = red[0];
    for (int j = threadIdx.x; j < cols; j += blockDim.x) o[j] = pr[j] * (dr[j] - dot);
}

// MaxPooling2D(2,2) backward: the gradient goes to the window's maximum (first one in row-major order on ties)
__global__ void maxpool2_bwd_kernel(const float* x, const float* dy, int B, int H, int W, int C, float* dx) {
    const int oH = H / 2, oW = W / 2;
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int xq = (int)(t % W); t /= W;
        const int yq = (int)(t % H);
        const int64_t b = t / H;
        const int oy = yq >> 1, ox = xq >> 1;
        float g = 0.f;
        if (oy < oH && ox < oW) {
            const float* base = x + ((b * H + 2 * oy) * W + 2 * ox) * C + c;
            const float v00 = base[0], v01 = base[C], v10 = base[(int64_t)W * C], v11 = base[(int64_t)W * C + C];
            const float m = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
            const int arg = v00 == m ? 0 : (v01 == m ? 1 : (v10 == m ? 2 : 3));
            if (arg == (yq & 1) * 2 + (xq & 1)) g = dy[((b * oH + oy) * oW + ox) * C + c];
        }
        dx[i] = g;
    }
}

// adjoint of the stride-2 pick (subsample2_kernel): dy [B, ceil(H/2), ceil(W/2), C] scattered back into a zero [B,H,W,C]
__global__ void zero_insert2_kernel(const float* dy, int B, int H, int W, int C, float* out) {
    const int oH = (H + 1) / 2, oW = (W + 1) / 2, offy = (H & 1) ? 0 : 1, offx = (W & 1) ? 0 : 1;
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int xq = (int)(t % W); t /= W;
        const int yq = (int)(t % H);
        const int64_t b = t / H;
        const int ry = yq - offy, rx = xq - offx;
        float g = 0.f;
        if (ry >= 0 && rx >= 0 && !(ry & 1) && !(rx & 1) && (ry >> 1) < oH && (rx >> 1) < oW) g = dy[((b * oH + (ry >> 1)) * oW + (rx >> 1)) * C + c];
        out[i] = g;
    }
}

// gradient of _spectral_loss (mean | |F(a)| - |F(b)| | over the (W, C) transform) w.r.t. a, scaled by `scale`:
//   da[w,c] = scale / count * Re( sum_{u,v} sgn(|Fa|-|Fb|)[u,v] * Fa[u,v]/|Fa[u,v]| * exp(+2 pi i (u w / W + v c / 3)) )
__global__ void __launch_bounds__(256) spectral_wc_bwd_kernel(const float* a, const float* b, int W, float scale, float* da) {
    extern __shared__ float sm[];
    float* tw = sm;                       // [W][2]  cos, sin of 2 pi k / W
    float* ga = tw + 2 * W;               // [W][3][2] channel DFT of a, later reused
    float* gb = ga + 6 * W;
    float* G = gb + 6 * W;                // [W][3][2] unit-phase * sign field in the (u, v) domain
    const int64_t row = blockIdx.x;
    const float* ra = a + row * W * 3;
    const float* rb = b + row * W * 3;
    const float c3 = -0.5f, s3 = 0.86602540378443864676f;
    for (int k = threadIdx.x; k < W; k += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)W, &s, &c);
        tw[2 * k] = (float)c; tw[2 * k + 1] = (float)s;
        for (int im = 0; im < 2; ++im) {
            const float* r = im ? rb : ra;
            float* g = im ? gb : ga;
            const float x0 = r[3 * k], x1 = r[3 * k + 1], x2 = r[3 * k + 2];
            g[6 * k + 0] = x0 + x1 + x2;        g[6 * k + 1] = 0.f;
            g[6 * k + 2] = x0 + c3 * (x1 + x2); g[6 * k + 3] = -s3 * (x1 - x2);
            g[6 * k + 4] = x0 + c3 * (x1 + x2); g[6 * k + 5] = s3 * (x1 - x2);
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 3 * W; o += blockDim.x) {
        const int u = o / 3, v = o - 3 * u;
        float are = 0.f, aim = 0.f, bre = 0.f, bim = 0.f;
        int k = 0;
        for (int w = 0; w < W; ++w) {
            const float tc = tw[2 * k], ts = -tw[2 * k + 1];                 // exp(-i theta)
            const float gar = ga[6 * w + 2 * v], gai = ga[6 * w + 2 * v + 1], gbr = gb[6 * w + 2 * v], gbi = gb[6 * w + 2 * v + 1];
            are += gar * tc - gai * ts; aim += gar * ts + gai * tc;
            bre += gbr * tc - gbi * ts; bim += gbr * ts + gbi * tc;
            k += u; if (k >= W) k -= W;
        }
        const float ma = sqrtf(are * are + aim * aim), mb = sqrtf(bre * bre + bim * bim);
        const float sg = ma > mb ? 1.f : (ma < mb ? -1.f : 0.f);
        const float inv = ma > 0.f ? sg / ma : 0.f;
        G[6 * u + 2 * v] = are * inv; G[6 * u + 2 * v + 1] = aim * inv;
    }
    __syncthreads();
    // inverse-direction sum over u per (w, v), then over v per channel
    for (int o = threadIdx.x; o < 3 * W; o += blockDim.x) {
        const int w = o / 3, c = o - 3 * w;
        float acc = 0.f;
        for (int v = 0; v < 3; ++v) {
            float re = 0.f, im = 0.f;
            int k = 0;
            for (int u = 0; u < W; ++u) {
                const float tc = tw[2 * k], ts = tw[2 * k + 1];              // exp(+i theta)
                const float gr = G[6 * u + 2 * v], gi = G[6 * u + 2 * v + 1];
                re += gr * tc - gi * ts; im += gr * ts + gi * tc;
                k += w; if (k >= W) k -= W;
            }
            // times exp(+2 pi i v c / 3), real part
            const int vc = (v * c) % 3;
            const float pc = vc == 0 ? 1.f : c3, ps = vc == 0 ? 0.f : (vc == 1 ? s3 : -s3);
            acc += re * pc - im * ps;
        }
        da[row * W * 3 + o] = scale * acc;
    }
}

unsigned grid_n(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

int wgrad_launch(sr_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int KS, float* dw, float* db, hipStream_t st) {
    return wgrad_launch_views(ctx, x, Cin, dy, Cout, B, H, W, Cin, Cout, KS, dw, db, st);
}

int eltwise_views_launch(sr_ctx* ctx, int op, const float* a, int64_t a_cs, const float* b, int64_t b_cs, float alpha, float beta, float* out, int64_t o_cs, int64_t npix, int C,
                         hipStream_t st) {
    if (npix <= 0 || C <= 0) return SR_OK;
    if (op < 0 || op > SR_ELT_SIGN_DIFF) return ctx->fail(SR_ERR_INVALID, "eltwise: unknown op");
    if (op != SR_ELT_AXPBY && op != SR_ELT_CLIP01 && !b) return ctx->fail(SR_ERR_INVALID, "eltwise: this op needs two operands");
    if (a_cs < C || o_cs < C || (b && b_cs < C)) return ctx->fail(SR_ERR_INVALID, "eltwise: a view is narrower than its channel count");
    hipLaunchKernelGGL(eltwise_view_kernel, dim3(grid_n(npix * C)), dim3(256), 0, st, op, a, a_cs, b, b_cs, alpha, beta, out, o_cs, npix, C);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int wgrad_launch_views(sr_ctx* ctx, const float* x, int64_t x_cs, const float* dy, int64_t dy_cs, int B, int H, int W, int Cin, int Cout, int KS, float* dw, float* db,
                       hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ctx->fail(SR_ERR_INVALID, "wgrad: empty tensor");
    if (x_cs < Cin || dy_cs < Cout) return ctx->fail(SR_ERR_INVALID, "wgrad: a view is narrower than its channel count");
    if ((x_cs != Cin || dy_cs != Cout) && KS != 3) return ctx->fail(SR_ERR_INVALID, "wgrad: channel-range views are built for 3x3 layers");
    if (KS < 1 || !(KS & 1) || KS > 15) return ctx->fail(SR_ERR_INVALID, "wgrad: odd kernel sizes up to 15 only");
    const int nci = (Cin + 31) / 32, nco = (Cout + 31) / 32, ntiles = KS * KS * nci * nco;
    const int64_t P = (int64_t)B * H * W;
    static const bool per_tap_env = getenv("SR355_WGRAD_PER_TAP") != nullptr;   // A/B switch (diagnostic): round 3's kernel for every layer (dense tensors only)
    const bool per_tap = per_tap_env && x_cs == Cin && dy_cs == Cout;
    if (KS == 3 && !per_tap) {
        const int tilesX = (W + WT_W - 1) / WT_W, tilesY = (H + WT_H - 1) / WT_H, ntile = B * tilesY * tilesX;
        // pixel splits: enough workgroups for ~1.5 per CU, each a whole number of 8 x 24 tiles
        int nsplit = (3 * ctx->cu_count() / 2 + nci * nco - 1) / (nci * nco);
        if (nsplit > ntile) nsplit = ntile;
        if (nsplit > 64) nsplit = 64;
        const int tiles_per_wg = (ntile + nsplit - 1) / nsplit;
        nsplit = (ntile + tiles_per_wg - 1) / tiles_per_wg;
        const size_t tile_bytes = (size_t)nsplit * ntiles * 1024 * sizeof(float);
        float* partial = static_cast<float*>(ctx->scratch(tile_bytes + (size_t)nsplit * nco * 32 * sizeof(double)));
        if (!partial) return SR_ERR_OOM;
        double* bpart = db ? reinterpret_cast<double*>(reinterpret_cast<char*>(partial) + tile_bytes) : nullptr;
        hipLaunchKernelGGL(wgrad3_tile_kernel, dim3(nci * nco, nsplit), dim3(256), 0, st, x, x_cs, dy, dy_cs, B, H, W, Cin, Cout, nci, nco, tiles_per_wg, tilesX, tilesY, partial, bpart);
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(ntiles, 4), dim3(256), 0, st, partial, ntiles, nsplit, Cin, Cout, nci, nco, dw, bpart, db);
        SR_HIP(ctx, hipGetLastError());
        return SR_OK;
    }
    int nsplit = (int)((P + WG_SPLIT_PIX - 1) / WG_SPLIT_PIX);
    if (nsplit > 64) nsplit = 64;
    if (nsplit < 1) nsplit = 1;
    float* partial = static_cast<float*>(ctx->scratch((size_t)nsplit * ntiles * 1024 * sizeof(float)));
    if (!partial) return SR_ERR_OOM;
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(ntiles, nsplit), dim3(256), 0, st, x, dy, B, H, W, Cin, Cout, KS, nci, nco, nsplit, partial);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(ntiles, 4), dim3(256), 0, st, partial, ntiles, nsplit, Cin, Cout, nci, nco, dw, (const double*)nullptr, (float*)nullptr);
    if (db) hipLaunchKernelGGL(colsum_kernel, dim3(Cout), dim3(256), 0, st, dy, P, Cout, db);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

// keras.optimizers.Adam (TF 2.10 optimizer_v2, dense update) over a flat fp32 bucket, in place:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  w = w - lr_t m / (sqrt(v) + eps),  lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) from the host.
// Every operation is rounded on its own (no fma contraction), in the order sr355.train.Adam's NumPy expression evaluates them, so the device
// update is bit for bit the host one (tests/test_train_gpu.py compares the two on one bucket).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr_t, float b1, float omb1, float b2, float omb2, float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = gscale == 1.f ? g[i] : __fmul_rn(g[i], gscale);
        const float mi = __fadd_rn(__fmul_rn(b1, m[i]), __fmul_rn(omb1, gi));
        const float vi = __fadd_rn(__fmul_rn(b2, v[i]), __fmul_rn(__fmul_rn(omb2, gi), gi));
        m[i] = mi;
        v[i] = vi;
        w[i] = __fsub_rn(w[i], __fdiv_rn(__fmul_rn(lr_t, mi), __fadd_rn(__fsqrt_rn(vi), eps)));
    }
}

int adam_launch(sr_ctx* ctx, float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float omb1, float b2, float omb2, float eps,
                float gscale, hipStream_t st) {
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "adam: empty bucket");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_n(n)), dim3(256), 0, st, w, g, m, v, n, lr_t, b1, omb1, b2, omb2, eps, gscale);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int eltwise_launch(sr_ctx* ctx, int op, const float* a, const float* b, float alpha, float beta, float* out, int64_t n, hipStream_t st) {
    if (n <= 0) return SR_OK;
    if (op < 0 || op > SR_ELT_SIGN_DIFF) return ctx->fail(SR_ERR_INVALID, "eltwise: unknown op");
    if (op != SR_ELT_AXPBY && op != SR_ELT_CLIP01 && !b) return ctx->fail(SR_ERR_INVALID, "eltwise: this op needs two operands");
    hipLaunchKernelGGL(eltwise_kernel, dim3(grid_n(n)), dim3(256), 0, st, op, a, b, alpha, beta, out, n);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int space_to_depth_launch(sr_ctx* ctx, const float* x, int B, int H, int W, int C, int r, float* y, hipStream_t st) {
    const int64_t n = (int64_t)B * H * W * r * r * C;
    if (n <= 0 || r < 1) return ctx->fail(SR_ERR_INVALID, "space_to_depth: bad shape");
    hipLaunchKernelGGL(space_to_depth_kernel, dim3(grid_n(n)), dim3(256), 0, st, x, B, H, W, C, r, y);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int matmul_launch(sr_ctx* ctx, const float* A, const float* B, float* C, int batch, int M, int N, int K, int tA, int tB, float alpha, hipStream_t st) {
    if (batch <= 0 || M <= 0 || N <= 0 || K <= 0) return ctx->fail(SR_ERR_INVALID, "matmul: empty operand");
    if (batch > 65535) return ctx->fail(SR_ERR_INVALID, "matmul: batch too large");
    hipLaunchKernelGGL(sgemm_kernel, dim3((N + 63) / 64, (M + 63) / 64, batch), dim3(256), 0, st, A, B, C, M, N, K, tA, tB, alpha, (int64_t)M * K, (int64_t)K * N,
                       (int64_t)M * N);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int softmax_rows_launch(sr_ctx* ctx, float* s, int64_t rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0 || rows >= (1ll << 31)) return ctx->fail(SR_ERR_INVALID, "softmax: bad shape");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int softmax_bwd_launch(sr_ctx* ctx, const float* p, const float* dp, float* ds, int64_t rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0 || rows >= (1ll << 31)) return ctx->fail(SR_ERR_INVALID, "softmax backward: bad shape");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, p, dp, ds, cols);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int maxpool2_bwd_launch(sr_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int C, float* dx, hipStream_t st) {
    const int64_t n = (int64_t)B * H * W * C;
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "maxpool backward: empty tensor");
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_n(n)), dim3(256), 0, st, x, dy, B, H, W, C, dx);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int zero_insert2_launch(sr_ctx* ctx, const float* dy, int B, int H, int W, int C, float* out, hipStream_t st) {
    const int64_t n = (int64_t)B * H * W * C;
    if (n <= 0) return ctx->fail(SR_ERR_INVALID, "zero insert: empty tensor");
    hipLaunchKernelGGL(zero_insert2_kernel, dim3(grid_n(n)), dim3(256), 0, st, dy, B, H, W, C, out);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int spectral_l1_bwd_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float scale, float* da, hipStream_t st) {
    if (C != 3) return ctx->fail(SR_ERR_INVALID, "spectral loss: built for 3 channels");
    constexpr int SPECTRAL_BWD_MAX_W = 160 * 1024 / 80;       // 80 bytes of LDS per pixel of width
    if (B <= 0 || H <= 0 || W <= 0) return ctx->fail(SR_ERR_INVALID, "spectral loss backward: bad shape");
    if (W > SPECTRAL_BWD_MAX_W) return ctx->fail(SR_ERR_INVALID, "spectral loss backward: image width " + std::to_string(W) + " exceeds " + std::to_string(SPECTRAL_BWD_MAX_W) + " (80 B of LDS per pixel of width)");
    const int64_t rows = (int64_t)B * H;
    const size_t lds = sizeof(float) * (size_t)(2 * W + 18 * W);
    auto kern = spectral_wc_bwd_kernel;
    if (lds > 48 * 1024) { if (int r2 = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)) return r2; }
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), lds, st, a, b, W, scale / (float)((double)rows * W * 3), da);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}
