// dense_fused.hip -- two consecutive 3x3 convs of an ESRGAN dense block (ESRGAN_model.py:212-254) in ONE persistent kernel.
//
// Why: layer by layer, a dense block moves ~2.3 KB per pixel through HBM (conv k re-reads every earlier feature) and the 3x3
// kernel of conv_rows.hip is bound by that stream (1.0x algorithmic traffic at 4.3 TB/s).  Here conv_a's output
// never leaves the CU before conv_b consumes it, and both convs share one staging of their common input chunks:
//   conv4 + conv5 (the pair that owns 65 % of a block's FLOP): 160 + 32 + 192 + 64 channel reads/writes -> 160 + 64.
//
// Shape of the computation (W = 48 exactly: three 16-pixel column groups; any H; any batch):
//   * a workgroup (12 waves, one workgroup per CU) owns a contiguous run of images and walks them top to bottom as ONE stream of rows,
//     8 rows per step, with a zero separator row between images (it is the bottom padding of one image and the top padding
//     of the next, so the MFMA loops never branch on image borders);
//   * line-buffer skew: in step s the first conv (layer 0) produces stream rows [8s, 8s+8), the second (layer 1) rows
//     [8s-1, 8s+7) -- layer 1's last input row is the row layer 0 finishes in the same step.  Layer 0's output lives in a
//     10-row ring in LDS (rows [8s-2, 8s+8)); nothing is recomputed, every layer computes exactly 8 rows per step;
//   * wave w owns row 8s+w of layer 0 and row 8s+w-1 of layer 1, all 48 columns, all output channels: the pixel fragment of
//     an input row feeds both layers and all three ky taps it takes part in (0.56 LDS fragment reads per MFMA);
//   * the external input (the dense block's concat buffer, row-blocked [B][H][C/32][W][32]: conv_common.h) is streamed one
//     32-channel chunk at a time: 11 rows x 3 KiB per chunk land in one of two (tail) / three (growth pairs) LDS buffers by LDS-DMA
//     (global_load_lds_dwordx4, source-side XOR swizzle so that ds_read_b128 is conflict free) while earlier chunks multiply;
//     weights go through a ring of LDS slots in (chunk, kx) granules, host-packed in the order they are used;
//   * roles: 8 compute waves (one stream row each; MFMAs, LDS reads, epilogues) + 4 loader waves (the whole DMA stream, one or two
//     granules / chunks ahead, counted vmcnt; they also hold part -- for the growth pairs all -- of the weights in their registers for
//     the whole kernel and put them into the slots with ds_write_b128); one s_barrier per granule (54 MFMAs per compute wave);
//   * output rows leave through an LDS transposition (whole cache lines instead of 64 16-byte fragments per store instruction).
// LDS: tail 2 x 33 + 30 + 3 x 18 KiB = 150 KiB, pairs 3 x 33 + 30 + 2 x 12 = 153 KiB -> one workgroup per CU, three waves per SIMD.
// What bounds the kernels (round 3): not the CU's memory port -- it delivers 59 B/clk from L2 and the loaders now issue a
// granule's pieces in ~0.5 k cycles -- but what this chip sustains for the loop's operand traffic: a skeleton of nothing but these MFMAs,
// LDS reads, DMA volume and one barrier per granule reaches 1.37-1.41 PFLOP/s at 1.75 GHz (tools/micro/mfma_ceiling.hip); the tail runs at
// 1.265.
//
// Epilogues: accumulators start at the bias; layer 0 = ReLU -> bf16 -> LDS ring (+ global when a later kernel needs it); layer 1 = either
// the same (growth conv) or the block's tail  alpha*(conv5 + b) + x [+ rrdb_in]  with both skips joining the accumulators on the matrix core
// as scaled-identity taps (x from the staged centre row, rrdb_in from two LDS-DMA'd pieces per ring granule), written to channels [0,64) of
// the next block's buffer.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "conv_common.h"

namespace {

struct ChainParams {
    const char* in; int in_nch;        // source concat buffer (row-blocked bf16), 32-channel chunks per pixel
    char* out; int out_nch;            // MODE 1: destination buffer of the block tail (channels [0, 64)); MODE 0: unused (outputs go into `in`)
    const char* so; int so_nch;        // MODE 1: the other skip tensor (64 channels at chunk 0 of a row-blocked buffer) or nullptr
    const char* w;                     // granule-ordered packed weights (pack_chain_weights)
    const float* bias;                 // [16*NB0 | 16*NB1] fp32
    const char* zero;                  // ZERO_PAGE_BYTES of zeros: DMA source of separator / out-of-stream rows (same chunk offsets as real rows)
    int B, H;
    int rows_per_wg; unsigned magic;   // stream rows (separators included) a workgroup owns; g / (H+1) == umulhi(g, magic) for every stream row index that occurs
    float alpha, xscale, oscale;       // MODE 1: out = alpha * (acc + bias + xscale * x + oscale * so); xscale = beta_x / alpha, oscale = beta_o / alpha, both exact in bf16
    int dbg_flags;                     // diagnostic builds only (env SR355_CHAIN_DBG_FLAGS): 1 = drop the tail's output stores, 2 = skip the external granules' MFMA bodies (timing experiments; results are wrong)
    unsigned long long* dbg;           // diagnostic builds only (sr_debug_set_chain_stamp_buffer): s_memtime stamps, [wg < 64][wave 0 / 5 / 8 / 11][granule < 64][4]
};

__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int ROWB = 3072;             // one LDS / HBM row of a chunk: 48 pixels x 64 B
constexpr int NSTG = 11;               // staged rows per chunk: stream rows [8s-2, 8s+9)
constexpr int STGB = NSTG * ROWB;
constexpr int WINR = 10;               // ring rows of layer 0's output: [8s-2, 8s+8)
constexpr int ZERO_PAGE_BYTES = sr_ctx::ZERO_PAGE_BYTES;

template <int NB0, int NB1, int MODE> struct ChainLds {
    static constexpr int WSLOT = (NB0 + NB1) * 3 * 1024;
    // Tail: two staging buffers (next chunk's rows fly during this chunk), three weight slots (the loaders' DMA runs two granules
    // ahead).  Growth pairs: their weights are register-resident in the loader waves (ds_write, one granule ahead: two slots), which
    // frees the room for a THIRD staging buffer -- rows are requested two chunks ahead, so a row piece has a whole chunk (~7 k
    // cycles) more than HBM's latency to land before anybody waits for it.
    static constexpr int NSB = MODE == 0 ? 3 : 2;
    static constexpr int NWS = MODE == 0 ? 2 : 3;
    static constexpr int BYTES = NSB * STGB + WINR * ROWB + NWS * WSLOT + (NB0 + NB1) * 16 * 4;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// EXT: external 32-channel chunks both convs read; NB0 / NB1: 16-cout blocks of layer 0 / layer 1.
// MODE 0: both layers are growth convs (ReLU) whose outputs go to chunks EXT and EXT+1 of the source buffer.
// MODE 1: layer 0 is a growth conv kept on chip only, layer 1 is the block tail (NB1 = 4).
#define CHAIN_STAMP(k) do { if (STAMP && blockIdx.x < 64 && G < 64 && (wave == 0 || wave == 5 || wave == 8 || wave == 11) && lane == 0)                \
        p.dbg[(((size_t)blockIdx.x * 4 + (wave == 0 ? 0 : wave == 5 ? 1 : wave == 8 ? 2 : 3)) * 64 + G) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// s_waitcnt vmcnt(N) with a compile-time N: everything but this wave's N youngest vector-memory operations has completed
template <int N> __device__ __forceinline__ void wait_imm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// What loader wave LW issues where, as compile-time functions of the granule position in a step (chain2_kernel's loader section).
template <int EXT, int NB0, int NB1, int MODE, int LW> struct LoaderPlan {
    static constexpr int NBT = NB0 + NB1, EXTG = 3 * EXT, NGR = 3 * (EXT + 1);
    static constexpr int RT_E = MODE == 1 ? 2 : 3;      // resident pieces per loader of an external granule's ceil(3 NBT / 4) = 5 (tail) / 3 (growth pair)
    static constexpr int RT_R = 2;                      // ... of a ring granule's ceil(3 NB1 / 4) = 3 / 2
    // rows of a chunk requested in the chunk's granule kx: {LW, LW + 4} | {8 + LW} (LW < 3) | {}
    static constexpr int nrows(int kx) { return kx == 0 ? 2 : kx == 1 ? (LW < 3 ? 1 : 0) : 0; }
    // row pieces requested at granule position i of a step (any integer: the pattern repeats every step)
    static constexpr int nst_at(int i) {
        i = (i % NGR + NGR) % NGR;
        return i < EXTG ? 3 * nrows(i % 3) : 0;
    }
    // weight pieces fetched by DMA (the non-resident ones) for the granule at position iw
    static constexpr int nwdma(int iw) {
        const bool ext = iw < EXTG;
        const int nw = ext ? NBT * 3 : NB1 * 3, rt = ext ? RT_E : RT_R;
        int n = 0;
        for (int t = 0; t < (nw + 3) / 4; ++t)
            if (LW + 4 * t < nw && t >= rt) ++n;
        return n;
    }
};

// (the same for the tail's 64 couts -- two 32-channel chunks of the row-blocked destination -- measured 0.3-0.5 % on the tail, inside the noise: the
// transposition through LDS stays there)
constexpr bool kDirectStores = true;   // growth convs (32 couts) store their rows straight from the accumulators (lane-pair swap: 16 contiguous bytes per lane)
constexpr int NCOMP = 8, NLOAD = 4;    // compute waves (one stream row each) + loader waves (LDS-DMA issue only), one loader per SIMD

// EXT: external 32-channel chunks both convs read; NB0 / NB1: 16-cout blocks of layer 0 / layer 1.
// MODE 0: both layers are growth convs (ReLU) whose outputs go to chunks EXT and EXT+1 of the source buffer.
// MODE 1: layer 0 is a growth conv kept on chip only, layer 1 is the block tail (NB1 = 4).
//
// Roles.  Four loader waves own the whole DMA stream and run two granules ahead (three weight slots, counted vmcnt): barrier -> issue ->
// wait for what the next barrier publishes -> barrier; the eight compute waves go barrier -> MFMAs -> barrier and touch vector memory only
// in the epilogues (and for the RRDB skip's two pieces per ring granule).  Round 2 read its stamps (200-300 cycles per DMA issue, the same
// with the MFMA bodies skipped) as a limit of the CU's L2 -> LDS path; round 3 measured that path alone (tools/micro/port_probe.hip:
// 17 cycles per 1 KiB piece from L2 with four loaders) and found the loaders' own ~45 scalar instructions per piece instead -- see the
// loader section below.  Measured in round 2 and not kept: 6 / 8 loaders (compute waves slower at 4 waves per SIMD and 128 registers),
// compute waves issuing the weight pieces themselves (-10 %).  The loaders also keep part (growth pairs: all) of the weights in their spare
// registers and write them into the slots with ds_write_b128: fewer L2 reads, and whole-line output stores through an LDS transposition
// (DESIGN.md 3.3 items 2 and 4 stand).
// SEAM (round 4): the 48-pixel rows hold TWO 24-pixel-wide images side by side (columns 0-23 | 24-47: patch_size_lr = 24, the reference's own training patch,
// ESRGAN_model.py:858 / constants.py:8, packed by api.hip) -- the only thing the kernel must know is that columns 23 and 24 are not neighbours: the kx = 0 fragment
// of column 24 and the kx = 2 fragment of column 23 are zero padding (lanes px = 8 resp. 7 of column group 1), exactly as columns -1 and 48 already are.
template <int EXT, int NB0, int NB1, int MODE, bool HAS_O, bool STAMP, bool SEAM = false>
__global__ void __launch_bounds__((NCOMP + NLOAD) * 64, (NCOMP + NLOAD + 3) / 4) chain2_kernel(ChainParams p) {
    using L = ChainLds<NB0, NB1, MODE>;
    constexpr int NBT = NB0 + NB1, WSLOT = L::WSLOT;
    constexpr int NGR = 3 * (EXT + 1);                 // granules per step: (chunk, kx) for the EXT external chunks + layer 0's own chunk
    constexpr int EXTG = 3 * EXT;
    constexpr int NTHR = (NCOMP + NLOAD) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const stg = smem;
    constexpr int NSB = L::NSB;
    char* const win = smem + NSB * STGB;
    char* const wr = win + WINR * ROWB;
    constexpr int NWS = L::NWS;
    float* const lbias = reinterpret_cast<float*>(wr + NWS * WSLOT);

    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, Hp1 = H + 1;
    // Round 3: a workgroup owns a contiguous range [R0, R1) of the GLOBAL stream of rows (all images, a zero separator row after each), not a
    // whole number of images: 882 patches on 256 CUs used to mean four images for 221 workgroups and none for 35 (what a rank sees at N = 8).
    // The range may start and end inside an image, so the local stream begins one row early (layer 0 recomputes row R0 - 1 for layer 1's
    // first own row) and ends one row late (layer 0's row R1 for layer 1's last); those two rows are computed, never stored -- the
    // neighbouring workgroups own and store them, with the same values.
    const int T = p.B * Hp1;                           // global stream rows
    const int R0 = blockIdx.x * p.rows_per_wg, R1 = min(T, R0 + p.rows_per_wg);
    if (R0 >= R1) return;                              // whole workgroup (uniform)
    const int Rs = R0 - 1;                             // global row of local stream row 0
    const int N = R1 - Rs + 1;                         // local stream rows layer 0 walks
    const int nsteps = (N + 1 + 7) >> 3;               // layer 1 lags one row

    // ---- one-time LDS state: the ring starts as zeros (rows above the first image), biases parked for the epilogues
    for (int u = tid; u < WINR * ROWB / 16; u += NTHR) *reinterpret_cast<f32x4*>(win + u * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < NBT * 16) lbias[tid] = p.bias[tid];
    // the compute waves read the biases (accumulator start values) before the first granule barrier: publish them now
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto row_of = [&](int g, int& img, int& y) -> bool {          // LOCAL stream row -> (image, row); false: separator / outside the stream
        const int gg = Rs + g;                                     // (local rows -2, -1 and those past N are real rows of the neighbours' ranges: the warm-up row needs them)
        if (gg < 0 || gg >= T) return false;
        img = (int)__umulhi((unsigned)gg, p.magic);
        y = gg - img * Hp1;
        return y < H;
    };
    auto own = [&](int g) { return Rs + g >= R0 && Rs + g < R1; };    // rows this workgroup stores
    int G = 0;
    auto dma = [&](const char* src, char* dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    if (wave >= NCOMP) {
        // =========================================================================================== loader waves
        // Round 3.  tools/micro/port_probe.hip (profiles/r03_port_probe.txt) measured what the CU's vector-memory path delivers into LDS
        // when the issuing loop is nothing but the DMA instructions: 59 B/clk from L2 (17 cycles per 1 KiB piece) with four loader
        // waves, 12 B/clk per CU from HBM with every CU streaming (the chip's 7 TB/s) and 25 B/clk with eight CUs -- beside eight MFMA
        // waves just the same.  Round 2's loaders spent ~200-250 cycles of SCALAR work per piece (row -> image by umulhi, 64-bit
        // multiplies, zero-page select, branches: ~45 instructions) and that, not the port, was the 54 k cycles per step the stamps
        // showed with the MFMA bodies skipped.  So: everything about a piece that does not change is computed once.
        //   * a loader owns whole rows of a staged chunk (three consecutive 1 KiB pieces: ONE address, immediate offsets 0 / 1024 / 2048
        //     on both the global and the LDS side) -- rows LW and LW + 4 in a chunk's first granule, row 8 + LW in its second, none in its
        //     third, so that no row is requested in the iteration whose barrier publishes it;
        //   * the base address of a row (image, row, separator or out of stream -> zero page) is computed once per STEP and row, in scalar
        //     registers, and serves every chunk of the step; the zero page is large enough to take the same chunk offset;
        //   * the loader's number is a compile-time constant (four copies of the code): piece lists, resident-weight indices and every
        //     s_waitcnt count are immediates; out-of-stream rows and the weights "after the last step" are issued all the same
        //     (zero page / valid memory, nobody reads them), so the counts never depend on the position in the stream.
        const int lwr = wave - NCOMP;
        // LDS-DMA source swizzle: lane i of a 1 KiB piece fills LDS slot i -> pixel i/4, slice position i%4, which must hold global slice
        // (i%4) ^ 2*bit2(pixel)
        const int lsrc = 64 * (lane >> 2) + 16 * ((lane & 3) ^ (2 * ((lane >> 4) & 1)));
        auto loader = [&](auto LWc) {
            constexpr int LW = decltype(LWc)::value;
            // base address of staged row j of step s2 (stream row 8 s2 - 2 + j), chunk 0, this lane's slice
            auto row_base = [&](int s2, int j) -> const char* {
                int img, y;
                const bool real = row_of(8 * s2 - 2 + j, img, y);
                return real ? p.in + ((int64_t)img * H + y) * p.in_nch * ROWB : p.zero;
            };
            // rows of a chunk this loader requests in the chunk's granule kx: {LW, LW + 4} | {8 + LW} (LW < 3) | {}
            using Plan = LoaderPlan<EXT, NB0, NB1, MODE, LW>;
            const char* rb_cur[3];
            const char* rb_nxt[3];
            auto bases_of = [&](int s2, const char* (&rb)[3]) {
                rb[0] = row_base(s2, LW);
                rb[1] = row_base(s2, LW + 4);
                rb[2] = row_base(s2, 8 + (LW < 3 ? LW : 2));
            };
            // one staged row = three pieces; rows 0..7 of a step are read by no later step (nt), rows 8..10 are the next step's halo
            auto stage_row = [&](auto Jc, const char* rb, int c1, char* sdst) {
                constexpr int j = decltype(Jc)::value;
                const char* src = rb + c1 * ROWB + lsrc;
                char* dst = sdst + j * ROWB;
                const auto gs = (const __attribute__((address_space(1))) void*)src;
                const auto ld = (__attribute__((address_space(3))) void*)dst;
                if constexpr (j < 8) {
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 2);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 2);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 2);
                } else {
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
                }
            };
            auto stage_rows = [&](auto KXc, const char* (&rb)[3], int c1, char* sdst) {
                constexpr int kx = decltype(KXc)::value;
                if constexpr (kx == 0) {
                    stage_row(std::integral_constant<int, LW>{}, rb[0], c1, sdst);
                    stage_row(std::integral_constant<int, LW + 4>{}, rb[1], c1, sdst);
                } else if constexpr (kx == 1 && LW < 3) {
                    stage_row(std::integral_constant<int, 8 + (LW < 3 ? LW : 0)>{}, rb[2], c1, sdst);
                }
            };
            // Resident weights.  Every 8-row step re-reads ALL weights: 54 of a tail chunk's 87 pieces.  A loader wave needs few registers
            // and owns 168, so each keeps the first RT_E (RT_R) of its pieces of every external (ring) granule in registers for the whole
            // kernel -- 132 KiB of the tail's 306 KiB, all of a growth pair's 90 / 126 KiB -- and writes them into the weight slot with
            // ds_write_b128 (the LDS write path, not the memory port) where the DMA version issues a load.
            constexpr int RT_E = Plan::RT_E, RT_R = Plan::RT_R;
            u32x4 wre[EXTG][RT_E], wrr[3][RT_R];
            auto wsrc_of = [&](int iw) { return p.w + (iw < EXTG ? iw * WSLOT : EXTG * WSLOT + (iw - EXTG) * (NB1 * 3 * 1024)); };
#pragma unroll
            for (int iw = 0; iw < EXTG; ++iw)
#pragma unroll
                for (int t = 0; t < RT_E; ++t)
                    if (LW + 4 * t < NBT * 3) wre[iw][t] = *reinterpret_cast<const u32x4*>(wsrc_of(iw) + (LW + 4 * t) * 1024 + lane * 16);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < RT_R; ++t)
                    if (LW + 4 * t < NB1 * 3) wrr[r][t] = *reinterpret_cast<const u32x4*>(wsrc_of(EXTG + r) + (LW + 4 * t) * 1024 + lane * 16);
            // weights of the granule at position IW of a step (running number Gw) -> slot Gw % NWS
            auto put_weights = [&](auto IW, int Gw) {
                constexpr int iw = decltype(IW)::value;
                constexpr bool ext = iw < EXTG;
                constexpr int nw = ext ? NBT * 3 : NB1 * 3, rt = ext ? RT_E : RT_R;
                const char* wq = p.w;
                asm volatile("" : "+s"(wq));                                 // recompute the piece addresses here: hoisted out of the step loop they cost two registers each
                const char* wsrc = wq + (ext ? iw * WSLOT : EXTG * WSLOT + (iw - EXTG) * (NB1 * 3 * 1024)) + lane * 16;
                char* wdst = wr + (Gw % NWS) * WSLOT;
#pragma unroll
                for (int t = 0; t < (nw + 3) / 4; ++t) {
                    const int k = LW + 4 * t;
                    if (k < nw) {
                        if (t < rt) {
                            if constexpr (ext) *reinterpret_cast<u32x4*>(wdst + k * 1024 + lane * 16) = wre[ext ? iw : 0][t < RT_E ? t : 0];
                            else *reinterpret_cast<u32x4*>(wdst + k * 1024 + lane * 16) = wrr[ext ? 0 : iw - EXTG][t < RT_R ? t : 0];
                        } else {
                            dma(wsrc + k * 1024, wdst + k * 1024);
                        }
                    }
                }
            };
            // prologue: weights of the first WL granules, the first SL external chunks of step 0; all of it is waited for before the first barrier
            constexpr int WL = NWS - 1;                                      // weights run WL granules ahead
            constexpr int SL = NSB - 1;                                      // rows run SL chunks ahead
            static_assert(MODE == 1 || (RT_E * NLOAD >= NBT * 3 && RT_R * NLOAD >= NB1 * 3), "growth pairs: every weight piece is resident (no weight DMA in the counted waits)");
            static_assert(SL <= EXT, "rows run at most one step ahead");
            static_for<WL>([&](auto I) { put_weights(I, decltype(I)::value); });
            bases_of(0, rb_cur);
            bases_of(1, rb_nxt);
            static_for<SL>([&](auto C0) {
                constexpr int c0 = decltype(C0)::value;
                stage_rows(std::integral_constant<int, 0>{}, rb_cur, c0, stg + c0 * STGB);
                stage_rows(std::integral_constant<int, 1>{}, rb_cur, c0, stg + c0 * STGB);
            });
            int nch = 0;
            for (int s = 0; s < nsteps; ++s) {
                static_for<NGR>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    CHAIN_STAMP(0);
                    // first iteration: the prologue has landed
                    if (G == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the resident pieces written in the previous iteration
                    __builtin_amdgcn_s_barrier();
                    CHAIN_STAMP(1);
                    constexpr int i2 = (i + WL) % NGR;                       // granule G+WL
                    // Order of issue: weights of granule G+WL, then this granule's rows.  vmcnt retires in order.  The barrier that ends this
                    // iteration publishes the weights of granule G+1 (issued first in the PREVIOUS iteration) and, after a chunk's third granule,
                    // all of the next chunk's rows (issued in its first two).  Tail: leave in flight this iteration's pieces and -- except in a
                    // third granule -- the previous iteration's rows (younger than its weights).  Growth pairs (no weight DMA, rows two
                    // chunks ahead): everything older than three iterations has landed.
                    put_weights(std::integral_constant<int, i2>{}, G + WL);
                    if constexpr (i < EXTG) {
                        constexpr int c = i / 3, kx = i - 3 * c;
                        constexpr int c1 = (c + SL) % EXT;
                        constexpr bool next_step = c + SL >= EXT;
                        char* sdst = stg + ((nch + SL) % NSB) * STGB;
                        stage_rows(std::integral_constant<int, kx>{}, next_step ? rb_nxt : rb_cur, c1, sdst);
                        if (kx == 2) ++nch;
                    }
                    CHAIN_STAMP(2);
                    if constexpr (MODE == 0) {
                        wait_imm<Plan::nst_at(i) + Plan::nst_at(i - 1) + Plan::nst_at(i - 2)>();
                    } else {
                        constexpr bool third = i < EXTG && i % 3 == 2;
                        wait_imm<Plan::nwdma(i2) + Plan::nst_at(i) + (third ? 0 : Plan::nst_at(i - 1))>();
                    }
                    CHAIN_STAMP(3);
                    ++G;
                });
                rb_cur[0] = rb_nxt[0]; rb_cur[1] = rb_nxt[1]; rb_cur[2] = rb_nxt[2];
                bases_of(s + 2, rb_nxt);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (lwr == 0) loader(std::integral_constant<int, 0>{});
        else if (lwr == 1) loader(std::integral_constant<int, 1>{});
        else if (lwr == 2) loader(std::integral_constant<int, 2>{});
        else loader(std::integral_constant<int, 3>{});
        return;
    }

    // =============================================================================================== compute waves
    // pixel fragment (B operand) of column group cg, tap kx at a row whose LDS base is R:  R + 1024*cg + offk[kx]
    //   column c = 16*cg + px + kx - 1, 64 B per pixel, 16-byte slice q at (q ^ 2*bit2(c)) -- conflict free for ds_read_b128
    int offk[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = px + kx - 1;
        offk[kx] = 64 * c + 16 * (q ^ (2 * ((c >> 2) & 1)));
    }
    const bool edge_l = px == 0, edge_r = px == 15;    // column -1 (kx 0, cg 0) and column 48 (kx 2, cg 2) are padding: fragment forced to zero
    const int off_l = edge_l ? offk[1] : offk[0];      // ... and read from a valid address
    const int off_r = edge_r ? offk[1] : offk[2];
    const bool seam_l = SEAM && px == 8, seam_r = SEAM && px == 7;      // column group 1: column 24's left neighbour / column 23's right neighbour belong to the other image

    f32x4 a0[NB0][3], a1[NB1][3];

    // Output rows leave through LDS.  In the MFMA result layout lane (px, q) holds 8 channels of pixel px, so neighbouring lanes are 64 B
    // apart and a 16-byte global store per lane reaches memory as 64 separate 16-byte writes: ~600 cycles per store instruction and wave
    // (stamps, the same with 8 workgroups on the chip as with 252), ~8 k cycles per step for the tail's 48.  Each 1 KiB piece (16 pixels x
    // 64 B) is therefore written to a private LDS slot in its memory order (slices XOR-swizzled by (px >> 1) & 3: conflict free) and read
    // back 16 B per lane in lane order, which stores -- and, for the RRDB skip, loads -- whole cache lines.  The slots live in the staging
    // buffer of the external chunk just finished, which nobody touches between the barrier after the last external granule and the one
    // that opens the next step (4 KiB per wave).
    // t_a: slot offset of channels [4q, 4q+4) of pixel px (channels [16+4q, ..) are at t_a ^ 32); t_line: slot offset of the piece's bytes
    // [16 l, 16 l + 16) for lane l.  Recomputed where they are used (once per step): kept live across the MFMA loops they were the values
    // hipcc spilled to scratch at the 168-register budget.
    auto transpose_offsets = [&](int& t_a, int& t_line) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int px_ = l & 15, q_ = l >> 4;
        t_a = px_ * 64 + 16 * ((q_ >> 1) ^ ((px_ >> 1) & 3)) + (q_ & 1) * 8;
        t_line = (l >> 2) * 64 + 16 * ((l & 3) ^ ((l >> 3) & 3));
    };

    // xscale * identity as MFMA A fragments: lane (cout i = lane & 15, k-quarter = lane >> 4) holds channels 8 (lane >> 4) + j; fragment h maps
    // channel 16 h + i of a 32-channel chunk onto cout i of a 16-cout block: its only non-zero element in this lane is j = 16 h + i - 8 q, if
    // that is in [0, 8).  Built where it is used (a few selects) rather than held in eight registers.  xscale (5 or 25 in this graph) is
    // exact in bf16: the host refuses anything that is not.
    auto xfold = [&](int h, float scale) {
        const bf16_t xs_b = (bf16_t)scale;
        int j0 = 16 * h + px - 8 * q;
        asm volatile("" : "+v"(j0));          // not loop invariant as far as hipcc can tell: hoisted, the fragments were spilled to scratch
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = j == j0 ? xs_b : (bf16_t)0.f;
        return f;
    };
    // Row-major walk of a granule: the three column-group fragments of one input row are read once and meet every weight fragment that
    // row pairs with (both layers, one ky each) before the next row is touched.
    // Round 3: a STATIC software pipeline.  Round 2's loop read each pair of weight fragments right in front of the six MFMAs that use
    // them (the compiler sinks LDS reads towards their use to save registers), so every six MFMAs met a full LDS latency: the stamps
    // show 108 MFMAs of a SIMD's two waves taking 2.2-2.4 k cycles (20-22 per MFMA, the pipe does 16).  Now the walk is a fixed list
    // of stages -- one weight fragment x three column groups each -- and the fragment of stage k + WDEPTH is requested before the
    // MFMAs of stage k; a sched_barrier after every stage keeps the compiler from undoing the distance.  The next row's pixel
    // fragments are requested at the first stage of the current row, as before.
    // (round 4, same-box A/B/A of the tail at 7056 patches: s_setprio 2 on the compute waves 1186.7 against 1186.6 / 1183.2 TFLOP/s, WDEPTH 4 1180.8, one
    //  sched_barrier per two stages 1181.2: nothing moves it -- profiles/r04_dense_variants_ab.txt)
    constexpr int WDEPTH = 3, NWREG = WDEPTH + 1;
    // one (chunk, kx) granule on an external chunk staged at `sb`: staged row j holds stream row 8s-2+j; layer 0 (row 8s+w) reads
    // j = w+1+ky, layer 1 (row 8s+w-1) reads j = w+ky.  Stage order: row d = 0..3: [layer 0, ky = d-1 (d >= 1)] [layer 1, ky = d (d <= 2)]
    struct ExtStage { int d, layer, n, frag, first; };
    auto ext_stage = [](int k) constexpr -> ExtStage {
        int i = 0;
        for (int d = 0; d < 4; ++d) {
            bool first = true;
            if (d >= 1)
                for (int n = 0; n < NB0; ++n, ++i, first = false)
                    if (i == k) return ExtStage{d, 0, n, (d - 1) * NB0 + n, first};
            if (d <= 2)
                for (int n = 0; n < NB1; ++n, ++i, first = false)
                    if (i == k) return ExtStage{d, 1, n, 3 * NB0 + d * NB1 + n, first};
        }
        return ExtStage{-1, 0, 0, 0, 0};
    };
    constexpr int NEXTST = 3 * (NB0 + NB1);
    auto ext_granule = [&](auto KXc, auto FOLDc, const char* sb, const char* ws) {
        constexpr int KX = decltype(KXc)::value;
        constexpr int FOLD = KX == 1 ? decltype(FOLDc)::value : -1;      // the centre tap carries the skip
        const char* rb = sb + wave * ROWB;
        auto load_row = [&](int d, bf16x8 (&x)[3]) {
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                const int off = (KX == 0 && cg == 0) ? off_l : (KX == 2 && cg == 2) ? off_r : offk[KX];
                x[cg] = *reinterpret_cast<const bf16x8*>(rb + d * ROWB + cg * 1024 + off);
                if ((KX == 0 && cg == 0 && edge_l) || (KX == 2 && cg == 2 && edge_r)) x[cg] = bf16x8{};
                if (SEAM && cg == 1 && ((KX == 0 && seam_l) || (KX == 2 && seam_r))) x[cg] = bf16x8{};
            }
        };
        auto ldw = [&](int frag) { return *reinterpret_cast<const bf16x8*>(ws + frag * 1024 + lane * 16); };
        bf16x8 xr[2][3], wq[NWREG];
        load_row(0, xr[0]);
        static_for<WDEPTH>([&](auto K) { wq[decltype(K)::value] = ldw(ext_stage(decltype(K)::value).frag); });
        static_for<NEXTST>([&](auto K) {
            constexpr int k = decltype(K)::value;
            constexpr ExtStage st = ext_stage(k);
            if constexpr (st.first && st.d < 3) load_row(st.d + 1, xr[(st.d + 1) & 1]);       // next row's fragments fly under this row's MFMAs
            if constexpr (k + WDEPTH < NEXTST) wq[(k + WDEPTH) % NWREG] = ldw(ext_stage(k + WDEPTH).frag);
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                if constexpr (st.layer == 0) a0[st.n][cg] = mma16(wq[k % NWREG], xr[st.d & 1][cg], a0[st.n][cg]);
                else a1[st.n][cg] = mma16(wq[k % NWREG], xr[st.d & 1][cg], a1[st.n][cg]);
            }
            // The block's own input x (channels [0, 64) = chunks 0, 1) is layer 1's skip: xscale * x(centre pixel) joins the accumulators of
            // cout blocks 2 FOLD, 2 FOLD + 1 as one more "tap" whose weight fragments are xscale times the identity (xfold[], built in
            // registers) against the centre row's fragments (staged row w + 1 = d 1, kx 1) -- six MFMAs where round 2 spent ~100 vector
            // instructions per wave and chunk (stamps: +1.0 k cycles on the two granules that carried it).
            if constexpr (FOLD >= 0 && st.d == 1 && st.layer == 1 && st.n == NB1 - 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x8 f = xfold(h, p.xscale);
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) a1[2 * FOLD + h][cg] = mma16(f, xr[1][cg], a1[2 * FOLD + h][cg]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // one kx granule of layer 1 on layer 0's output: ring rows r1-1, r1, r1+1 with r1 = 8s+w-1
    constexpr int NRINGST = 3 * NB1;
    auto ring_granule = [&](auto KXc, int s, const char* ws) {
        constexpr int KX = decltype(KXc)::value;
        auto load_row = [&](int ky, bf16x8 (&x)[3]) {
            const char* rb = win + ((8 * s + wave - 2 + ky + 2 * WINR) % WINR) * ROWB;
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                const int off = (KX == 0 && cg == 0) ? off_l : (KX == 2 && cg == 2) ? off_r : offk[KX];
                x[cg] = *reinterpret_cast<const bf16x8*>(rb + cg * 1024 + off);
                if ((KX == 0 && cg == 0 && edge_l) || (KX == 2 && cg == 2 && edge_r)) x[cg] = bf16x8{};
                if (SEAM && cg == 1 && ((KX == 0 && seam_l) || (KX == 2 && seam_r))) x[cg] = bf16x8{};
            }
        };
        auto ldw = [&](int frag) { return *reinterpret_cast<const bf16x8*>(ws + frag * 1024 + lane * 16); };
        bf16x8 xr[2][3], wq[NWREG];
        load_row(0, xr[0]);
        static_for<WDEPTH>([&](auto K) { wq[decltype(K)::value] = ldw(decltype(K)::value); });
        static_for<NRINGST>([&](auto K) {
            constexpr int k = decltype(K)::value;            // stage k: ky = k / NB1, cout block n = k % NB1, fragment k
            constexpr int ky = k / NB1, n = k % NB1;
            if constexpr (n == 0 && ky < 2) load_row(ky + 1, xr[(ky + 1) & 1]);
            if constexpr (k + WDEPTH < NRINGST) wq[(k + WDEPTH) % NWREG] = ldw(k + WDEPTH);
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) a1[n][cg] = mma16(wq[k % NWREG], xr[ky & 1][cg], a1[n][cg]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // Granule boundary of a compute wave: its LDS reads / writes of the finished granule are done, then the barrier that publishes the
    // loaders' pieces.  No vmcnt here: a compute wave issues no DMA, and its epilogue stores drain on their own.
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    int nch = 0;
    auto chunk = [&](auto FOLDc, int& Gr) {
        const char* sb = stg + (nch % NSB) * STGB;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++Gr) {
            CHAIN_STAMP(0);
            sync();
            CHAIN_STAMP(1);
            CHAIN_STAMP(2);
            const char* ws = wr + (Gr % NWS) * WSLOT;
            if (STAMP && (p.dbg_flags & 2)) { /* timing experiment: loaders alone */ }
            else if (kx == 0) ext_granule(std::integral_constant<int, 0>{}, FOLDc, sb, ws);
            else if (kx == 1) ext_granule(std::integral_constant<int, 1>{}, FOLDc, sb, ws);
            else ext_granule(std::integral_constant<int, 2>{}, FOLDc, sb, ws);
            CHAIN_STAMP(3);
        }
        ++nch;
    };
    for (int s = 0; s < nsteps; ++s) {
        // accumulators start at the bias (the epilogues then have no bias add)
#pragma unroll
        for (int n = 0; n < NB0; ++n) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + n * 16 + 4 * q);
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) a0[n][cg] = b;
        }
#pragma unroll
        for (int n = 0; n < NB1; ++n) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + (NB0 + n) * 16 + 4 * q);
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) a1[n][cg] = b;
        }
        if constexpr (MODE == 1) {
            static_assert(MODE == 0 || EXT >= 2, "the tail's skip spans chunks 0 and 1");
            chunk(std::integral_constant<int, 0>{}, G);
            chunk(std::integral_constant<int, 1>{}, G);
#pragma nounroll
            for (int c = 2; c < EXT; ++c) chunk(std::integral_constant<int, -1>{}, G);
        } else {
#pragma nounroll
            for (int c = 0; c < EXT; ++c) chunk(std::integral_constant<int, -1>{}, G);
        }
        // ---- layer 0 epilogue: ReLU -> bf16 -> ring row (8s+w) mod 10 (zeros on separator / out-of-stream rows)
        {
            const int g0 = 8 * s + wave;
            int img, y;
            const bool real = row_of(g0, img, y);
            char* wrow = win + (g0 % WINR) * ROWB;
            char* grow = real && own(g0) && MODE == 0 ? const_cast<char*>(p.in) + (((int64_t)img * H + y) * p.in_nch + EXT) * ROWB : nullptr;
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                f32x4 v[NB0];
                bf16x4 ob[NB0];
#pragma unroll
                for (int n = 0; n < NB0; ++n) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[n][e] = real ? fmaxf(a0[n][cg][e], 0.f) : 0.f;
                    const bf16x4 o = {(bf16_t)v[n][0], (bf16_t)v[n][1], (bf16_t)v[n][2], (bf16_t)v[n][3]};
                    ob[n] = o;
                    const int slice = n * 2 + (q >> 1);
                    *reinterpret_cast<bf16x4*>(wrow + cg * 1024 + 64 * px + 16 * (slice ^ (2 * ((px >> 2) & 1))) + (q & 1) * 8) = o;
                }
                if constexpr (MODE == 0 && NB0 == 2 && kDirectStores) {
                    // 32 couts: lanes q, q ^ 1 trade halves and every lane stores 16 contiguous bytes, a column group one contiguous KiB -- no read-back of the ring row
                    if (grow) {
                        const u32x2 au = __builtin_bit_cast(u32x2, ob[0]), cu = __builtin_bit_cast(u32x2, ob[NB0 - 1]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], cu[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], cu[1], false, false);
                        const u32x4 ov = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(grow + cg * 1024 + 64 * px + 2 * ((q & 1) * 16 + 4 * (q & ~1))));
                    }
                }
            }
            if (MODE == 0 && grow && !(NB0 == 2 && kDirectStores)) {
                // the ring row just written IS the row's memory image (64 B per pixel, slices swizzled by 2 * bit2(pixel)): read it back in
                // lane order and store whole lines
                static_assert(MODE == 1 || NB0 == 2, "growth convs have 32 output channels");
                asm volatile("" ::: "memory");
#pragma unroll
                for (int cg = 0; cg < 3; ++cg) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wrow + cg * 1024 + (lane >> 2) * 64 + 16 * ((lane & 3) ^ (2 * ((lane >> 4) & 1))));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(grow + cg * 1024 + lane * 16));
                }
            }
        }
        char* const tb = stg + ((nch + NSB - 1) % NSB) * STGB + wave * 4096;       // transposition slots (free staging buffer, see above)
        // The other skip tensor of an RRDB's last block (beta_o * so) joins layer 1's accumulators as (beta_o / alpha) * so while layer 1 runs
        // on the ring, the same way x does: column group kx's two 1 KiB pieces (chunks 0 and 1 of the wave's row) come by LDS-DMA in the
        // staged pixel layout into the wave's transposition slots 0 / 1 -- requested right behind the barrier of ring granule kx -- and meet
        // oscale * identity fragments in four MFMAs after that granule's own.  (Round 2 held the
        // two pieces in eight registers through the granule, transposed them through LDS and spent ~50 vector instructions per granule.)
        const char* sob = nullptr;
        if (MODE == 1 && HAS_O) {
            int img, y;
            if (row_of(8 * s + wave - 1, img, y) && own(8 * s + wave - 1)) sob = p.so + ((int64_t)img * H + y) * p.so_nch * ROWB;
        }
        auto so_fetch = [&](int cg) {
            if (MODE == 1 && HAS_O && sob) {
                int l = lane;
                asm volatile("" : "+v"(l));
                const int ls = 64 * (l >> 2) + 16 * ((l & 3) ^ (2 * ((l >> 4) & 1)));      // the loaders' source swizzle
                dma(sob + cg * 1024 + ls, tb);
                dma(sob + ROWB + cg * 1024 + ls, tb + 1024);
            }
        };
        // ---- layer 1 on layer 0's output
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++G) {
            CHAIN_STAMP(0);
            sync();
            CHAIN_STAMP(1);
            // behind the barrier, every wave: the slots sit in the staging buffer of the last external chunk, which other waves read until
            // they have passed the barrier of the first ring granule
            so_fetch(kx);
            CHAIN_STAMP(2);
            const char* ws = wr + (G % NWS) * WSLOT;
            if (kx == 0) ring_granule(std::integral_constant<int, 0>{}, s, ws);
            else if (kx == 1) ring_granule(std::integral_constant<int, 1>{}, s, ws);
            else ring_granule(std::integral_constant<int, 2>{}, s, ws);
            if (MODE == 1 && HAS_O && sob) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's own two pieces (and its long-finished output stores)
                int l = lane;
                asm volatile("" : "+v"(l));
                const int pxl = l & 15, ql = l >> 4;
                const int offc = 64 * pxl + 16 * (ql ^ (2 * ((pxl >> 2) & 1)));          // centre-tap pixel fragment inside a 1 KiB piece
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const bf16x8 xo = *reinterpret_cast<const bf16x8*>(tb + c2 * 1024 + offc);
#pragma unroll
                    for (int h = 0; h < 2; ++h) a1[2 * c2 + h][kx] = mma16(xfold(h, p.oscale), xo, a1[2 * c2 + h][kx]);
                }
            }
            CHAIN_STAMP(3);
        }
        // ---- layer 1 epilogue (row 8s+w-1): ReLU | * alpha -> bf16 -> transposition slot -> whole-line stores
        {
            int img, y;
            const bool real = row_of(8 * s + wave - 1, img, y) && own(8 * s + wave - 1);
            if (real && !(STAMP && (p.dbg_flags & 1))) {
                const int64_t rowi = (int64_t)img * H + y;
                char* const grow = MODE == 0 ? const_cast<char*>(p.in) + (rowi * p.in_nch + EXT + 1) * ROWB : p.out + rowi * p.out_nch * ROWB;
                const float alpha = p.alpha;
                int t_a, t_line;
                transpose_offsets(t_a, t_line);
                if constexpr (MODE == 0 && NB1 == 2 && kDirectStores) {
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) {
                        bf16x4 ob[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            ob[n] = bf16x4{(bf16_t)fmaxf(a1[n][cg][0], 0.f), (bf16_t)fmaxf(a1[n][cg][1], 0.f), (bf16_t)fmaxf(a1[n][cg][2], 0.f), (bf16_t)fmaxf(a1[n][cg][3], 0.f)};
                        const u32x2 au = __builtin_bit_cast(u32x2, ob[0]), cu = __builtin_bit_cast(u32x2, ob[1]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], cu[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], cu[1], false, false);
                        const u32x4 ov = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(grow + cg * 1024 + 64 * px + 2 * ((q & 1) * 16 + 4 * (q & ~1))));
                    }
                } else
#pragma unroll
                for (int cg = 0; cg < 3; ++cg)
#pragma unroll
                    for (int h = 0; h < NB1 / 2; ++h) {
                        char* const slot = tb + ((2 * cg + h) & 3) * 1024;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int n = 2 * h + u;
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = MODE == 0 ? fmaxf(a1[n][cg][e], 0.f) : alpha * a1[n][cg][e];
                            *reinterpret_cast<bf16x4*>(slot + (t_a ^ (32 * u))) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        }
                        asm volatile("" ::: "memory");                 // the read-back below is of other lanes' writes: keep the order
                        const u32x4 line = *reinterpret_cast<const u32x4*>(slot + t_line);
                        __builtin_nontemporal_store(line, reinterpret_cast<u32x4*>(grow + h * ROWB + cg * 1024 + lane * 16));
                    }
            }
        }
    }
}

// ==================================================================================================================================
// conv1 of a dense block (64 -> 32 channels, ESRGAN_model.py:230-233) as a streaming kernel of its own.
//
// It is the one HBM-bound conv of the block (192 FLOP per byte against a ridge of 312): on the tile kernel a workgroup lives ~11 us for ~1 us of
// MFMAs -- load a tile, wait, multiply, store -- and four of them per CU keep 4.7 TB/s in flight.  Here the rows of a workgroup's range of the
// global row stream (as chain2_kernel: images top to bottom, a zero separator row after each) flow through FOUR staging buffers -- two 32-channel
// chunks x two 8-row steps, 120 KiB -- filled by four loader waves 1.5 steps ahead of the eight compute waves, while the conv's whole 36 KiB of weights
// sit in LDS for the life of the kernel (no weight slots, no ring, one barrier per chunk).  Output rows leave straight from the accumulators: with 32
// couts a lane pair's permlane16_swap gives every lane 16 contiguous bytes and a column group's 64 lanes one contiguous KiB of the row-blocked row.
// Weights: the row-sliding kernel's packed layout (conv.hip: [chunk][tap = ky * 3 + kx][cout block][lane][8]) as it is.
struct Conv1Params {
    const char* in; int in_nch;        // row-blocked concat buffer: reads chunks 0, 1, writes chunk 2
    const char* w;                     // 36 KiB packed weights (ConvWeights of the 64 -> 32 conv)
    const float* bias;                 // [32]
    const char* zero;                  // zero page
    int B, H;
    int rows_per_wg; unsigned magic;
};

constexpr int C1_NSTG = 10;            // staged rows per chunk and step: stream rows [8s - 1, 8s + 9)
constexpr int C1_STGB = C1_NSTG * ROWB;
constexpr int C1_NSB = 4;
constexpr int C1_WBYTES = 2 * 9 * 2 * 1024;
constexpr int C1_LDS = C1_NSB * C1_STGB + C1_WBYTES + 32 * 4;
static_assert(C1_LDS <= 160 * 1024, "LDS budget");

template <bool SEAM>
__global__ void __launch_bounds__((NCOMP + NLOAD) * 64, (NCOMP + NLOAD + 3) / 4) conv1_stream_kernel(Conv1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const stg = smem;
    char* const lw = smem + C1_NSB * C1_STGB;
    float* const lbias = reinterpret_cast<float*>(lw + C1_WBYTES);
    constexpr int NTHR = (NCOMP + NLOAD) * 64;
    const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, Hp1 = H + 1;
    const int T = p.B * Hp1;
    const int R0 = blockIdx.x * p.rows_per_wg, R1 = min(T, R0 + p.rows_per_wg);
    if (R0 >= R1) return;
    const int nsteps = (R1 - R0 + 7) >> 3;                         // local stream row g = global row R0 + g
    const int nchunks = 2 * nsteps;                                // (step, chunk) units in order

    // one-time: the whole conv's weights and the biases
    for (int u = tid; u < C1_WBYTES / 16; u += NTHR) *reinterpret_cast<f32x4*>(lw + u * 16) = *reinterpret_cast<const f32x4*>(p.w + u * 16);
    if (tid < 32) lbias[tid] = p.bias[tid];

    auto row_of = [&](int g, int& img, int& y) -> bool {          // local stream row -> (image, row); false: separator / outside the stream
        const int gg = R0 + g;
        if (gg < 0 || gg >= T) return false;
        img = (int)__umulhi((unsigned)gg, p.magic);
        y = gg - img * Hp1;
        return y < H;
    };
    if (wave >= NCOMP) {
        // ------------------------------------------------------------------------------------------------ loader waves
        // unit k = (step k / 2, chunk k % 2) goes to buffer k % 4.  Loader LW stages rows LW, LW + 4 and (LW < 2) 8 + LW of every unit: three
        // 1 KiB pieces per row from one address.  Schedule per barrier k (the barrier that lets the compute waves start unit k): issue unit
        // k + 3 (its buffer was released by the barrier before: unit k - 1 is done), then wait until unit k + 1 has landed = all but this
        // loader's pieces of units k + 2 and k + 3.
        const int lwr = wave - NCOMP;
        const int lsrc = 64 * (lane >> 2) + 16 * ((lane & 3) ^ (2 * ((lane >> 4) & 1)));
        auto loader = [&](auto LWc) {
            constexpr int LW = decltype(LWc)::value;
            constexpr int NROWS = LW < 2 ? 3 : 2, NP = 3 * NROWS;
            auto stage_unit = [&](int k) {                         // every unit is issued, also those past the end (zero page: nobody reads them)
                const int s2 = k >> 1, c = k & 1;
                char* const sdst = stg + (k % C1_NSB) * C1_STGB;
                static_for<NROWS>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value == 0 ? LW : decltype(Jc)::value == 1 ? LW + 4 : 8 + LW;
                    int img, y;
                    const bool real = k < nchunks && row_of(8 * s2 - 1 + j, img, y);
                    const char* src = (real ? p.in + (((int64_t)img * H + y) * p.in_nch + c) * ROWB : p.zero) + lsrc;
                    const auto gs = (const __attribute__((address_space(1))) void*)src;
                    const auto ld = (__attribute__((address_space(3))) void*)(sdst + j * ROWB);
                    // (the nt policy for the rows no later step re-reads, which keeps the fused pairs' halo rows in L2, measured 2 % slower here)
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
                    __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
                });
            };
            stage_unit(0); stage_unit(1); stage_unit(2);
            for (int k = 0; k < nchunks; ++k) {
                if (k == 0) wait_imm<2 * NP>();                    // unit 0 has landed (units 1, 2 may fly)
                __builtin_amdgcn_s_barrier();                      // compute may start unit k; unit k - 1's buffer is free
                stage_unit(k + 3);
                wait_imm<2 * NP>();                                // unit k + 1 has landed
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        // the one-time LDS writes above are published by the first barrier (lgkmcnt first)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lwr == 0) loader(std::integral_constant<int, 0>{});
        else if (lwr == 1) loader(std::integral_constant<int, 1>{});
        else if (lwr == 2) loader(std::integral_constant<int, 2>{});
        else loader(std::integral_constant<int, 3>{});
        return;
    }
    // ---------------------------------------------------------------------------------------------------- compute waves
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own share of the weight copy
    int offk[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = px + kx - 1;
        offk[kx] = 64 * c + 16 * (q ^ (2 * ((c >> 2) & 1)));
    }
    const bool edge_l = px == 0, edge_r = px == 15;
    const int off_l = edge_l ? offk[1] : offk[0];
    const int off_r = edge_r ? offk[1] : offk[2];
    typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
    for (int s = 0; s < nsteps; ++s) {
        f32x4 acc[2][3];
        // (the biases were written before the first barrier below)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = 2 * s + c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c == 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + n * 16 + 4 * q);
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) acc[n][cg] = b;
                }
            }
            const char* const sb = stg + (k % C1_NSB) * C1_STGB + wave * ROWB;       // staged row j = w + ky
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8 wf[3][2];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int n = 0; n < 2; ++n) wf[ky][n] = *reinterpret_cast<const bf16x8*>(lw + (((c * 9) + ky * 3 + kx) * 2 + n) * 1024 + lane * 16);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int cg = 0; cg < 3; ++cg) {
                        const int off = (kx == 0 && cg == 0) ? off_l : (kx == 2 && cg == 2) ? off_r : offk[kx];
                        bf16x8 x = *reinterpret_cast<const bf16x8*>(sb + ky * ROWB + cg * 1024 + off);
                        if ((kx == 0 && cg == 0 && edge_l) || (kx == 2 && cg == 2 && edge_r)) x = bf16x8{};
                        if (SEAM && cg == 1 && ((kx == 0 && px == 8) || (kx == 2 && px == 7))) x = bf16x8{};      // two 24-pixel images per row: see chain2_kernel
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc[n][cg] = mma16(wf[ky][n], x, acc[n][cg]);
                    }
                }
            }
        }
        // ---- epilogue: ReLU -> bf16 -> chunk 2 of the row, 16 contiguous bytes per lane (lanes q, q ^ 1 trade halves)
        int img, y;
        const int g = 8 * s + wave;
        if (g < R1 - R0 && row_of(g, img, y)) {
            char* const grow = const_cast<char*>(p.in) + (((int64_t)img * H + y) * p.in_nch + 2) * ROWB;
            const int lane_b = 64 * px + 2 * ((q & 1) * 16 + 4 * (q & ~1));
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                bf16x4 o[2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    o[n] = bf16x4{(bf16_t)fmaxf(acc[n][cg][0], 0.f), (bf16_t)fmaxf(acc[n][cg][1], 0.f), (bf16_t)fmaxf(acc[n][cg][2], 0.f), (bf16_t)fmaxf(acc[n][cg][3], 0.f)};
                const u32x2c au = __builtin_bit_cast(u32x2c, o[0]), cu = __builtin_bit_cast(u32x2c, o[1]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(au[0], cu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(au[1], cu[1], false, false);
                const u32x4 ov = {(unsigned)s0[0], (unsigned)s1[0], (unsigned)s0[1], (unsigned)s1[1]};
                __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(grow + cg * 1024 + lane_b));
            }
        }
    }
}

uint16_t bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <int EXT, int NB0, int NB1, int MODE>
int launch_chain(sr_ctx* ctx, const ChainParams& p, bool has_o, int nwg, bool seam, hipStream_t st) {
    constexpr int lds = ChainLds<NB0, NB1, MODE>::BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (seam) {                                   // two 24-pixel-wide images per row
        if (has_o) {
            auto k = chain2_kernel<EXT, NB0, NB1, MODE, true, false, true>;
            if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), lds)) return rc;
            hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), lds, st, p);
        } else {
            auto k = chain2_kernel<EXT, NB0, NB1, MODE, false, false, true>;
            if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), lds)) return rc;
            hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), lds, st, p);
        }
    } else if (p.dbg) {                                  // diagnostic stamped variant (never in production)
        auto k = chain2_kernel<EXT, NB0, NB1, MODE, MODE == 1, true>;
        if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), lds)) return rc;
        ChainParams q = p;
        if (MODE == 1 && !has_o) { q.so = p.in; q.so_nch = p.in_nch; q.oscale = 0.f; }   // the stamped build always carries the skip loads
        hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), lds, st, q);
    } else if (has_o) {
        auto k = chain2_kernel<EXT, NB0, NB1, MODE, true, false>;
        if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), lds)) return rc;
        hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), lds, st, p);
    } else {
        auto k = chain2_kernel<EXT, NB0, NB1, MODE, false, false>;
        if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), lds)) return rc;
        hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), lds, st, p);
    }
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
// Weights of the pair (conv_a: Cin_a = 32*ext -> 16*nb0 couts, conv_b: Cin_b = 32*(ext+1) -> 16*nb1 couts), HWIO fp32, packed in
// the order the kernel consumes them: for each external chunk c and kx: [a: ky x cout block][b: ky x cout block] (1 KiB MFMA
// A-fragments, lane l element j = W[ky][kx][32c + 8(l>>4) + j][16 blk + (l&15)]), then for conv_b's last chunk (= conv_a's
// output) and kx: [b: ky x cout block].
int chain_pack_weights(sr_ctx* ctx, const float* wa, const float* ba, const float* wb, const float* bb, int ext, int nb0, int nb1, ChainWeights* out) {
    const int cin_a = 32 * ext, cin_b = 32 * (ext + 1), cout_a = 16 * nb0, cout_b = 16 * nb1;
    const size_t nfrag = (size_t)3 * ext * 3 * (nb0 + nb1) + (size_t)3 * 3 * nb1;
    std::vector<uint16_t> host(nfrag * 512);
    size_t idx = 0;
    auto frag = [&](const float* w, int cin, int cout, int c, int ky, int kx, int blk) {
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j)
                host[idx++] = bf16_host(w[((size_t)(ky * 3 + kx) * cin + 32 * c + 8 * (l >> 4) + j) * cout + 16 * blk + (l & 15)]);
    };
    for (int c = 0; c < ext; ++c)
        for (int kx = 0; kx < 3; ++kx) {
            for (int ky = 0; ky < 3; ++ky) for (int n = 0; n < nb0; ++n) frag(wa, cin_a, cout_a, c, ky, kx, n);
            for (int ky = 0; ky < 3; ++ky) for (int n = 0; n < nb1; ++n) frag(wb, cin_b, cout_b, c, ky, kx, n);
        }
    for (int kx = 0; kx < 3; ++kx)
        for (int ky = 0; ky < 3; ++ky) for (int n = 0; n < nb1; ++n) frag(wb, cin_b, cout_b, ext, ky, kx, n);
    ChainWeights cw;
    cw.ext = ext; cw.nb0 = nb0; cw.nb1 = nb1;
    cw.bytes = host.size() * 2;
    cw.w = ctx->dalloc(cw.bytes);
    if (!cw.w) return SR_ERR_OOM;
    cw.bias = static_cast<float*>(ctx->dalloc(sizeof(float) * (cout_a + cout_b)));
    if (!cw.bias) { ctx->dfree(cw.w); return SR_ERR_OOM; }
    std::vector<float> hb(cout_a + cout_b, 0.f);
    if (ba) for (int i = 0; i < cout_a; ++i) hb[i] = ba[i];
    if (bb) for (int i = 0; i < cout_b; ++i) hb[cout_a + i] = bb[i];
    SR_HIP(ctx, hipMemcpy(cw.w, host.data(), cw.bytes, hipMemcpyHostToDevice));
    SR_HIP(ctx, hipMemcpy(cw.bias, hb.data(), sizeof(float) * hb.size(), hipMemcpyHostToDevice));
    *out = cw;
    return SR_OK;
}

void chain_free_weights(sr_ctx* ctx, ChainWeights* w) {
    if (w->w) ctx->dfree(w->w);
    if (w->bias) ctx->dfree(w->bias);
    w->w = nullptr; w->bias = nullptr;
}

// conv1 of a dense block on the streaming kernel: `in` is the block's row-blocked concat buffer (reads channels [0, 64), writes [64, 96)),
// `w` the conv's ordinary packed weights (row-sliding layout, 64 -> 32)
bool conv1_stream_supported(const ConvWeights& w, const TensorView& in, int W) {
    return w.w != nullptr && w.rows && w.dtype == SR_DTYPE_BF16 && w.KS == 3 && w.Cin == 64 && w.Cout == 32 && w.NT == 2 && w.nchunks == 2 && W == 48 && in.blk &&
           in.coff == 0 && in.cs % 32 == 0 && in.cs >= 96;
}

int conv1_stream_launch(sr_ctx* ctx, const ConvWeights& w, TensorView in, int B, int H, int W, hipStream_t st, bool seam) {
    if (!conv1_stream_supported(w, in, W)) return ctx->fail(SR_ERR_INVALID, "streaming conv1: needs a 64 -> 32 bf16 3x3 conv on a 48-pixel-wide row-blocked buffer");
    if (B <= 0 || H <= 0) return ctx->fail(SR_ERR_INVALID, "streaming conv1: empty tensor");
    if (!ctx->zero_page) return ctx->fail(SR_ERR_STATE, "context has no zero page");      // sr_init allocates and clears it
    int ncu = ctx->cu_count();
    if (ctx->chain_max_wgs > 0 && ctx->chain_max_wgs < ncu) ncu = ctx->chain_max_wgs;
    Conv1Params p;
    p.in = static_cast<const char*>(in.p); p.in_nch = (int)(in.cs / 32);
    p.w = static_cast<const char*>(w.w); p.bias = w.bias; p.zero = static_cast<const char*>(ctx->zero_page);
    p.B = B; p.H = H;
    const int64_t T = (int64_t)B * (H + 1);
    if ((T + 16) * (int64_t)(H + 1) * (H + 1) >= (1ll << 32)) return ctx->fail(SR_ERR_INVALID, "streaming conv1: stream too long");
    const int nwg_target = (int)std::max<int64_t>(1, std::min<int64_t>(ncu, (T + 15) / 16));
    p.rows_per_wg = (int)((T + nwg_target - 1) / nwg_target);
    const int nwg = (int)((T + p.rows_per_wg - 1) / p.rows_per_wg);
    p.magic = (unsigned)(((1ull << 32) + (unsigned)H) / (unsigned)(H + 1));
    int rec = -1;
    if (ctx->prof) {
        const double px = (double)B * H * W;
        rec = ctx->prof_open("dense_conv1_stream<bf16,64->32>", 2.0 * px * 9.0 * 64 * 32, px * 2.0 * (64 + 32), st);
    }
    auto k = seam ? conv1_stream_kernel<true> : conv1_stream_kernel<false>;
    if (int rc = ctx->ensure_dyn_lds(reinterpret_cast<const void*>(k), C1_LDS)) return rc;
    hipLaunchKernelGGL(k, dim3(nwg), dim3((NCOMP + NLOAD) * 64), C1_LDS, st, p);
    ctx->prof_close(rec, st);
    SR_HIP(ctx, hipGetLastError());
    return SR_OK;
}

bool chain_supported(const ChainWeights& w, const TensorView& in, int W) {
    return w.w != nullptr && W == 48 && in.blk && in.coff == 0 && in.cs % 32 == 0 && in.cs >= 32 * (w.ext + (w.nb1 == 4 ? 1 : 2));
}

// in: the dense block's row-blocked concat buffer.  tail (nb1 == 4): out = alpha*(conv_b + bias) + beta_x * x + beta_o * skip_o into channels
// [0,64) of `out` (x = channels [0,64) of `in`).  Otherwise both convs are growth convs writing chunks ext and ext+1 of `in`.
int chain_launch(sr_ctx* ctx, const ChainWeights& w, TensorView in, int B, int H, int W, TensorView out, TensorView skip_o, float alpha,
                 float beta_x, float beta_o, hipStream_t st, bool seam) {
    if (!chain_supported(w, in, W)) return ctx->fail(SR_ERR_INVALID, "fused dense-block pair: needs W == 48 and a row-blocked source buffer");
    if (B <= 0 || H <= 0) return ctx->fail(SR_ERR_INVALID, "fused dense-block pair: empty tensor");
    const bool tail = w.nb1 == 4;
    if (tail && (!out.p || !out.blk || out.coff != 0 || out.cs % 32 != 0 || alpha == 0.f)) return ctx->fail(SR_ERR_INVALID, "fused dense-block tail: bad destination view");
    if (skip_o.p && (!skip_o.blk || skip_o.coff != 0 || skip_o.cs % 32 != 0)) return ctx->fail(SR_ERR_INVALID, "fused dense-block tail: bad skip view");
    // separator / out-of-stream rows are staged from the zero page with the same chunk offset as real rows: (EXT - 1) * 3 KiB + one 3 KiB row
    static_assert(ZERO_PAGE_BYTES >= 6 * ROWB, "zero page covers every chunk offset");
    if (!ctx->zero_page) return ctx->fail(SR_ERR_STATE, "context has no zero page");      // sr_init allocates and clears it
    int ncu = ctx->num_cus;
    if (ncu <= 0) {
        hipDeviceProp_t prop;
        SR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ncu = ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (ctx->chain_max_wgs > 0 && ctx->chain_max_wgs < ncu) ncu = ctx->chain_max_wgs;      // test hook: several images per workgroup at small batches
    ChainParams p;
    p.in = static_cast<const char*>(in.p); p.in_nch = (int)(in.cs / 32);
    p.out = static_cast<char*>(const_cast<void*>(out.p)); p.out_nch = (int)(out.cs / 32);
    p.so = static_cast<const char*>(skip_o.p); p.so_nch = (int)(skip_o.cs / 32);
    p.w = static_cast<const char*>(w.w); p.bias = w.bias; p.zero = static_cast<const char*>(ctx->zero_page);
    p.B = B; p.H = H;
    // rows of the global stream per workgroup: all CUs busy whatever the batch, but no fewer than 24 rows each (two of a range's rows are
    // recomputed for its neighbours)
    const int64_t T = (int64_t)B * (H + 1);
    if ((T + 16) * (int64_t)(H + 1) * (H + 1) >= (1ll << 32)) return ctx->fail(SR_ERR_INVALID, "fused dense-block pair: stream too long");
    const int nwg_target = (int)std::max<int64_t>(1, std::min<int64_t>(ncu, (T + 23) / 24));
    p.rows_per_wg = (int)((T + nwg_target - 1) / nwg_target);
    static const bool whole_images = getenv("SR355_CHAIN_WHOLE_IMAGES") != nullptr;       // A/B switch (diagnostic): round 2's partition, ranges = whole images
    if (whole_images) p.rows_per_wg = ((B + ncu - 1) / ncu) * (H + 1);
    const int nwg = (int)((T + p.rows_per_wg - 1) / p.rows_per_wg);
    p.magic = (unsigned)(((1ull << 32) + (unsigned)H) / (unsigned)(H + 1));                 // ceil(2^32 / (H+1)): exact quotient for g (H+1)^2 < 2^32 (checked above)
    p.alpha = alpha; p.xscale = tail ? beta_x / alpha : 0.f; p.oscale = tail && skip_o.p ? beta_o / alpha : 0.f;
    for (float sc : {p.xscale, p.oscale}) {   // the skips join the accumulators as scale * identity MFMA fragments in bf16: only exactly representable ratios (5 and 25 here)
        uint32_t u = (uint32_t)bf16_host(sc) << 16;
        float back;
        memcpy(&back, &u, 4);
        if (back != sc) return ctx->fail(SR_ERR_INVALID, "fused dense-block tail: beta / alpha is not exactly representable in bf16");
    }
    p.dbg = ctx->chain_stamp_buf;
    { const char* f = getenv("SR355_CHAIN_DBG_FLAGS"); p.dbg_flags = f ? atoi(f) : 0; }
    if (p.dbg && ctx->chain_stamp_skip >= 0 && ctx->chain_stamp_skip-- != 0) p.dbg = nullptr;   // stamp one chosen launch only
    if (p.dbg && ctx->chain_stamp_skip == -1 && getenv("SR355_CHAIN_STAMP_SKIP")) ctx->chain_stamp_buf = nullptr;
    int rec = -1;
    if (ctx->prof) {
        const double px = (double)B * H * W;
        const double cin_a = 32.0 * w.ext, cin_b = 32.0 * (w.ext + 1), ca = 16.0 * w.nb0, cb = 16.0 * w.nb1;
        // algorithmic bytes: the shared input once, what leaves the chip, the skips
        double bytes = px * 2.0 * (cin_a + (tail ? cb : ca + cb));   // the shared input once + what leaves the chip (the tail's x skip is part of the input)
        if (tail && skip_o.p) bytes += px * 2.0 * 64.0;
        rec = ctx->prof_open(tail ? "dense_tail_fused<bf16,conv4+conv5>" : "dense_pair_fused<bf16>", 2.0 * px * 9.0 * (cin_a * ca + cin_b * cb), bytes, st);
    }
    int rc;
    const bool has_o = skip_o.p != nullptr;
    if (tail && w.ext == 5 && w.nb0 == 2) rc = launch_chain<5, 2, 4, 1>(ctx, p, has_o, nwg, seam, st);
    else if (!tail && w.ext == 2 && w.nb0 == 2 && w.nb1 == 2) rc = launch_chain<2, 2, 2, 0>(ctx, p, false, nwg, seam, st);
    else if (!tail && w.ext == 3 && w.nb0 == 2 && w.nb1 == 2) rc = launch_chain<3, 2, 2, 0>(ctx, p, false, nwg, seam, st);
    else rc = ctx->fail(SR_ERR_INVALID, "fused dense-block pair: shape not instantiated");
    ctx->prof_close(rec, st);
    return rc;
}
