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
    // the 4 waves interleave pixel pairs
    for (int64_t p = p0 + 2 * wave; p < p1; p += 8) {
        const int64_t pp = p + k;
        float a = 0.f, bv = 0.f;
        if (pp < p1) {
            const int xq = (int)(pp % W);
            const int64_t r = pp / W;
            const int yq = (int)(r % H);
            const int sy = yq + ky - pad, sx = xq + kx - pad;
            if (ci_ok && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) a = x[((r - yq + sy) * W + sx) * Cin + ci];
            if (co_ok) bv = dy[pp * Cout + co];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
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

__global__ void wgrad_finish_kernel(const float* partial, int ntiles, int nsplit, int Cin, int Cout, int nci, int nco, float* dw) {
    // one thread per element of one 32x32 tile; dw is HWIO [tap][Cin][Cout]
    const int tile = blockIdx.x;
    int t = tile;
    const int cob = t % nco; t /= nco;
    const int cib = t % nci;
    const int tap = t / nci;
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
        const int ci = cib * 32 + e / 32, co = cob * 32 + (e & 31);
        if (ci >= Cin || co >= Cout) continue;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += partial[((size_t)sp * ntiles + tile) * 1024 + e];
        dw[((size_t)tap * Cin + ci) * Cout + co] = s;
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

unsigned grid_n(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

}  // namespace

int wgrad_launch(sr_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int KS, float* dw, float* db, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ctx->fail(SR_ERR_INVALID, "wgrad: empty tensor");
    if (KS < 1 || !(KS & 1) || KS > 15) return ctx->fail(SR_ERR_INVALID, "wgrad: odd kernel sizes up to 15 only");
    const int nci = (Cin + 31) / 32, nco = (Cout + 31) / 32, ntiles = KS * KS * nci * nco;
    const int64_t P = (int64_t)B * H * W;
    int nsplit = (int)((P + WG_SPLIT_PIX - 1) / WG_SPLIT_PIX);
    if (nsplit > 64) nsplit = 64;
    if (nsplit < 1) nsplit = 1;
    float* partial = static_cast<float*>(ctx->scratch((size_t)nsplit * ntiles * 1024 * sizeof(float)));
    if (!partial) return SR_ERR_OOM;
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3(ntiles, nsplit), dim3(256), 0, st, x, dy, B, H, W, Cin, Cout, KS, nci, nco, nsplit, partial);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(ntiles), dim3(256), 0, st, partial, ntiles, nsplit, Cin, Cout, nci, nco, dw);
    if (db) hipLaunchKernelGGL(colsum_kernel, dim3(Cout), dim3(256), 0, st, dy, P, Cout, db);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

int eltwise_launch(sr_ctx* ctx, int op, const float* a, const float* b, float alpha, float beta, float* out, int64_t n, hipStream_t st) {
    if (n <= 0) return SR_OK;
    if (op < 0 || op > SR_ELT_CLIP01) return ctx->fail(SR_ERR_INVALID, "eltwise: unknown op");
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
