// conv_pw.hip -- bf16 1x1 convolution (attention projections, SRCNN conv2) as a streaming GEMM on v_mfma_f32_16x16x32_bf16.
//
// A 1x1 conv reads every input byte once and writes every output byte once: it is a pure HBM stream with Cin*Cout/(Cin+Cout)
// FLOP per byte (55 for 64 -> 48).  No LDS, no barriers: the whole weight matrix (Cin/32 x Cout/16 fragments of 4 VGPRs, at most
// 16 of them) and the biases live in registers for the lifetime of a wave, which walks 16-pixel row segments (memory-level parallelism comes from the 16 waves per CU);
// the pixel fragment of a segment is the MFMA's B operand straight from global memory (lane = pixel, 16 B of channels), and the
// epilogue is the row kernels' (conv_rows_epi.h): a lane owns 4 consecutive couts of a pixel, pairs of lanes trade halves for
// 16-byte stores where the channel count allows.  NHWC views only.
#include "conv_rows_epi.h"

namespace {

using namespace convk;

// NCH 32-channel chunks x NB16 16-cout blocks of weight fragments held in registers (NCH, NB16 <= 4).  The segment loop is
// instantiated per epilogue kind (conv_rows_epi.h) so that only one variant's loop invariants are live.
template <int NB16, int NCH, int KIND>
__device__ __forceinline__ void pw_loop(const ConvParams& p, int xblocks, int nunits) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, q = lane >> 4;
    const int H = p.H, W = p.W, in_cs = (int)p.in_cs;
    bf16x8 wf[NCH][NB16];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int n = 0; n < NB16; ++n) wf[c][n] = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)(c * NB16 + n) * 64 + lane) * 16);
    f32x4 biasv[NB16];
#pragma unroll
    for (int n = 0; n < NB16; ++n) biasv[n] = *reinterpret_cast<const f32x4*>(p.bias + n * 16 + 4 * q);
    const bf16_t* in = reinterpret_cast<const bf16_t*>(p.in) + p.in_coff + q * 8;
    const int stride = gridDim.x * 4;
    for (int u = blockIdx.x * 4 + wave; u < nunits; u += stride) {
        const int row = u / xblocks, x0 = (u - row * xblocks) * 16;
        const int x = min(x0 + px, W - 1);
        const bf16_t* src = in + ((int64_t)row * W + x) * in_cs;
        bf16x8 xf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) xf[c] = *reinterpret_cast<const bf16x8*>(src + c * 32);
        f32x4 acc[1][NB16];
#pragma unroll
        for (int n = 0; n < NB16; ++n) acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int n = 0; n < NB16; ++n) acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][n], xf[c], acc[0][n], 0, 0, 0);
        const int b = row / H, y = row - b * H;
        rows_epilogue_as<NB16, 1, KIND>(p, acc, biasv, b, y, x0, 0, 0, px, q);
    }
}

template <int NB16, int NCH>
__global__ void __launch_bounds__(256, 4) conv1_pw_kernel(ConvParams p, int xblocks, int nunits) {
    switch (rows_epilogue_kind<NB16>(p)) {
        case -1: pw_loop<NB16, NCH, -1>(p, xblocks, nunits); break;
        case 0: pw_loop<NB16, NCH, 0>(p, xblocks, nunits); break;
        case 1: pw_loop<NB16, NCH, 1>(p, xblocks, nunits); break;
        case 2: pw_loop<NB16, NCH, 2>(p, xblocks, nunits); break;
        case 3: pw_loop<NB16, NCH, 3>(p, xblocks, nunits); break;
        case 4: pw_loop<NB16, NCH, 4>(p, xblocks, nunits); break;
        default: pw_loop<NB16, NCH, 5>(p, xblocks, nunits); break;
    }
}

template <int NB16, int NCH>
int launch_pw(sr_ctx* ctx, const ConvParams& p, hipStream_t st) {
    const int xblocks = (p.W + 15) / 16;
    const int64_t nunits = (int64_t)p.B * p.H * xblocks;
    if (nunits >= (1ll << 31)) return ctx->fail(SR_ERR_INVALID, "conv_pw: too many row segments for one launch");
    const int64_t wgs = (nunits + 3) / 4;                            // 4 waves, one segment per wave and pass
    const int grid = (int)(wgs < 256 * 8 * 8 ? wgs : 256 * 8 * 8);   // ~8 passes per wave on a full chip: weights and biases are loaded once per wave
    hipLaunchKernelGGL((conv1_pw_kernel<NB16, NCH>), dim3(grid > 0 ? grid : 1), dim3(256), 0, st, p, xblocks, (int)nunits);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

}  // namespace

int conv_pw_launch(sr_ctx* ctx, const ConvWeights& w, const convk::ConvParams& p, hipStream_t st) {
#define SR_PW(NB, NC) if (w.NT == NB && w.nchunks == NC) return launch_pw<NB, NC>(ctx, p, st)
    SR_PW(1, 1); SR_PW(1, 2); SR_PW(1, 3); SR_PW(1, 4);
    SR_PW(2, 1); SR_PW(2, 2); SR_PW(2, 3); SR_PW(2, 4);
    SR_PW(3, 1); SR_PW(3, 2); SR_PW(3, 3);
    SR_PW(4, 1);
#undef SR_PW
    return ctx->fail(SR_ERR_INVALID, "conv_pw: unsupported (cout blocks, cin chunks) combination");
}
