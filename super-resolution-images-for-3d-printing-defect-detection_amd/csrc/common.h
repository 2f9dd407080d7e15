// common.h -- internal declarations shared by the libsr355 translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/sr355.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// context: one per GPU.  Owns nothing but its tracked allocations and two timing events.
// ---------------------------------------------------------------------------------------------
struct sr_ctx {
    int device = 0;
    std::string err;
    int64_t cur_bytes = 0, peak_bytes = 0;
    std::unordered_map<void*, size_t> allocs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;

    void* scratch_buf = nullptr;
    size_t scratch_cap = 0;
    unsigned long long* stamp_buf = nullptr;   // diagnostic: when set, conv3_rows runs its stamped variant
    int64_t stamp_cap = 0;                     // ... bytes behind stamp_buf: a launch that would write beyond is refused
    int chain_stamp_skip = -1;                        // diagnostic: >= 0 -> only the launch after that many fused launches is stamped (env SR355_CHAIN_STAMP_SKIP)
    unsigned long long* chain_stamp_buf = nullptr;   // diagnostic: when set, the fused dense-block kernels run their stamped variant

    // per-launch HIP-event timing of the hot kernels (sr_profile_begin/_end)
    struct ProfRec { int name; hipEvent_t e0, e1; double flops, bytes; };
    bool prof = false;
    std::vector<std::string> prof_names;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    // returns record index or -1 (profiling off); e0 is recorded on `st` before the launch
    int prof_open(const std::string& name, double flops, double bytes, hipStream_t st);
    void prof_close(int rec, hipStream_t st);

    // kernels whose dynamic-LDS ceiling has been raised on THIS context's device (hipFuncSetAttribute is per device)
    static constexpr int MAX_LDS_BYTES = 160 * 1024;          // LDS of one gfx950 CU
    std::unordered_map<const void*, int> lds_attr_done;      // kernel -> largest dynamic-LDS size its attribute has been raised to
    int ensure_dyn_lds(const void* kernel, int bytes);
    void* tab_buf = nullptr; size_t tab_cap = 0;   // tap tables of sr_resize (stream-ordered reuse)
    struct Arena { void* p = nullptr; size_t cap = 0; };
    Arena attn_kn;                // attention: per-key-group largest key norm (stream-ordered reuse)
    Arena dev_w, dev_b, dev_x;    // sr_conv2d_dev: packed weights / padded bias / padded input of the call in flight (stream-ordered reuse)
    void* arena(Arena& a, size_t bytes, hipStream_t st);   // grow-only; growing waits for `st` first
    // sr_conv_prepack: packed fp32 weights of a list of conv uses, written by one launch and found again by conv_pack_weights_dev
    struct PackKey {
        const void* w; const void* b; int K, Cin, Cout, rot;
        bool operator==(const PackKey& o) const { return w == o.w && b == o.b && K == o.K && Cin == o.Cin && Cout == o.Cout && rot == o.rot; }
    };
    struct PackKeyHash {
        size_t operator()(const PackKey& k) const {
            return std::hash<const void*>()(k.w) ^ (std::hash<const void*>()(k.b) * 31u) ^ ((size_t)k.rot << 1) ^ ((size_t)k.Cin << 20) ^ ((size_t)k.Cout << 36) ^ ((size_t)k.K << 52);
        }
    };
    std::unordered_map<PackKey, std::pair<void*, float*>, PackKeyHash> pack_cache;
    Arena pack_w, pack_b, pack_tab;
    std::vector<char> pack_tab_host;     // the job table as last uploaded
    void* pack_tab_dev = nullptr;
    static constexpr int ZERO_PAGE_BYTES = 32768;
    void* zero_page = nullptr;    // ZERO_PAGE_BYTES of zeros (DMA source of padding rows / halo pixels in dense_fused.hip, conv_stream.hip): allocated and cleared
                                  // synchronously in sr_init, so that no launch on any stream can see it before it is zero
    int num_cus = 0;
    int cu_count();               // compute units of the device (queried once)
    int chain_mask = 511;          // bit 0: fuse conv4+conv5 of a dense block, bit 1: fuse conv2+conv3, bit 2: fold the generator's RGB conv into final_conv1, bit 3: SelfAttention's f / g / h projections in the producing conv's epilogue bit 4: batches of small images (VGG16 block 5) packed in a CellGrid, bit 5: conv1 of a dense block on the streaming kernel, bit 6: a 2x2 max-pool inside the conv in front of it, bit 7: 64-input-channel 3x3 convs on the persistent kernel of conv_stream.hip, bit 8: SRCNN's 1x1 conv inside the 9x9 head's epilogue (sr_debug_set_fused; default all)
    int chain_max_wgs = 0;        // test hook: cap the persistent grid so that small batches still give several images per workgroup
    int64_t alloc_cap = 0;        // test hook (sr_debug_set_alloc_cap): dalloc fails once cur_bytes would exceed it; 0 = none

    void* dalloc(size_t bytes);   // nullptr on failure (err set)
    void dfree(void* p);
    void* scratch(size_t bytes);  // reduction scratch; stream-ordered reuse (one stream per ctx at a time)
    int fail(int code, const std::string& msg) { err = msg; return code; }
};

// Every ABI entry that allocates or launches binds the calling thread to the context's device first and puts the
// previous device back on return (the current device is per-thread state the caller -- torch -- may have changed).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const sr_ctx* c) {
        if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define SR_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return (ctx)->fail(SR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

static inline int dtype_size(int dt) { return dt == SR_DTYPE_F32 ? 4 : (dt == SR_DTYPE_BF16 ? 2 : 1); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// conv: packed weights + launch
// ---------------------------------------------------------------------------------------------
struct ConvWeights {          // device-resident, MFMA-fragment-ordered (see conv.hip header)
    void* w = nullptr;        // packed kernel
    float* bias = nullptr;    // [CoutP] fp32, zero padded
    int dtype = SR_DTYPE_BF16;
    int KS = 3, Cin = 0, Cout = 0;
    int CinP = 0, CoutP = 0;  // padded sizes the kernel iterates over
    int thin = 0;             // 1: Cin fits one 16-byte slice per pixel (taps paired in a k-group)
    int KGPT = 2;             // wide: k-groups per tap per stage (2 -> 64 B of channels, 4 -> 128 B)
    int NT = 1;               // 32-wide cout blocks per workgroup (rows variant: 16-wide blocks, NB16)
    int rows = 0;             // 1: bf16 3x3 row-sliding kernel (conv_rows.hip) and its weight layout
    int few = 0;              // 1: fp32, <= 4 couts: VALU kernel (conv.hip conv_fewcout_f32_kernel), weights [tap][CinP][4]
    int pw = 0;               // 1: bf16 1x1 streaming kernel (conv_pw.hip), same fragment layout, NT = all cout blocks
    int nchunks = 1;          // wide: Cin stages;  thin: unused
    size_t bytes = 0;
};

struct TensorView {           // NHWC view with a channel stride/offset (elements)
    const void* p = nullptr;
    int64_t cs = 0;           // channels per pixel in the underlying buffer
    int coff = 0;             // first channel of this view
    int blk = 0;              // 1: the buffer is row-blocked, [B][H][cs/32][W][32] (conv_common.h); cs, coff multiples of 32
};

// Fused RGB tail (conv_rows.hip, rows_fuse2): the 3x3 conv to c2 <= 3 channels that follows a 64-cout 3x3 conv, folded into that conv's epilogue
struct RgbTailWeights {
    void* a = nullptr;        // four 1 KiB MFMA A-fragments [tap-channel block][channel half], bf16
    float* bias = nullptr;    // [3] fp32, zero padded
    int c2 = 0;
};

// Fused 1x1 projection (conv_rows.hip, rows_epilogue_proj): a 1x1 conv of a 64-channel conv output to 16 * nblk <= 48 channels
struct ProjWeights {
    void* a = nullptr;        // [nblk][2] 1 KiB MFMA A-fragments, bf16
    float* bias = nullptr;    // [16 * nblk] fp32
    int nblk = 0;
};

// Fused 1x1 conv behind an fp32 thin conv (conv.hip, conv_thin_kernel): SRCNN's conv2d_1 (96 -> 32, ReLU) computed in conv2d's epilogue from the
// accumulators, so that the 96-channel fp32 tensor never reaches HBM (SRCNN_model.py:48-53)
struct Pw2Weights {
    float* a = nullptr;       // [NT = cin / 32][16][64 lanes] fp32: the A operand of the (n, i) step of the 32x32x2 fp32 MFMA chain, 0 beyond cout2
    float* bias = nullptr;    // [32] fp32, zero padded
    int cin = 0, cout = 0, act = 0;
};

// Packed layout of a batch of SMALL images (api.hip: VGG16 block 5).  A 6 x 6 image uses 19 % of the 12 x 16 output tile the 64-cout kernel
// issues MFMAs for.  Image b instead sits in cell (b / gx, b % gx) of a grid of ch x cw = (h + 1) x (w + 1) cells whose last row / column is a
// ZERO separator (the bottom / right padding of one image and the top / left padding of the next), and the whole batch is ONE image of Hv x Wv
// pixels to the 3x3 kernel (gx = 2: 12 of 16 tile columns, 6 of 7 rows used).  The separators are zero because the buffer is zeroed when the
// layout is adopted and nothing writes them: the conv epilogue skips them (ConvEpilogue::cell_h / cell_w), the pool writes image pixels only.
// gx = 0: plain NHWC [B, H, W, C].
struct CellGrid {
    int gx = 0, ch = 0, cw = 0, Hv = 0, Wv = 0;
    bool operator==(const CellGrid& o) const { return gx == o.gx && ch == o.ch && cw == o.cw && Hv == o.Hv && Wv == o.Wv; }
};

struct ConvEpilogue {
    TensorView pool_out;                  // conv_rows, 64 couts per workgroup, no skips: store only the 2x2 / stride-2 maximum of the output (keras MaxPooling2D), NHWC bf16 ...
    CellGrid pool_grid;                   // ... or packed in this CellGrid
    int cell_h = 0, cell_w = 0;           // conv_rows: do not store output rows y with y % cell_h == cell_h - 1 / columns x with x % cell_w == cell_w - 1 (CellGrid separators)
    const ProjWeights* pj = nullptr;      // conv_rows, 64 couts per output pixel, NHWC output: also write the 1x1 projection of the output to pj_out
    TensorView pj_out;                    // NHWC bf16 view at the conv's output resolution, >= 16 * nblk channels from coff
    const Pw2Weights* pw2 = nullptr;      // fp32 thin conv owning all its couts in one workgroup, no skips: store act2(W2 . act(conv) + b2) (<= 32 channels) instead of the conv's own output
    const RgbTailWeights* f2 = nullptr;   // conv_rows, 64 couts, no skips: do not store this conv's output, write the following conv's partial sums to f2_part
    float* f2_part = nullptr;             // rgbtail_partial_bytes(B, H, W) bytes
    int act = SR_ACT_LINEAR;
    float alpha = 1.f;
    TensorView skip1, skip2;  // same dtype as the conv's compute dtype; p == nullptr -> unused
    float beta1 = 0.f, beta2 = 0.f;
    int clip01 = 0;
    int d2s_r = 1;            // depth_to_space block (1 = none)
    int out_f32 = 0;          // write fp32 even when computing in bf16
    int allow_splitk = 0;     // fp32 3x3: small problems may split K across the waves of a workgroup (conv_wide_sk_kernel) -- another summation order than the tile kernel's,
                              // so only the training entry points set it: a model's forward must give the same bits whatever the batch it is called with (Keras predict's chunking)
};

// host: pack HWIO fp32 weights (+bias) for the device.  Returns SR_OK or error (ctx->err set).
// the same packing on the device (fp32 only): `d_hwio` is a device HWIO tensor, [K,K,Cin,Cout], or with rot = 1 the kernel whose
// 180-degree-rotated, channel-swapped form is wanted ([K,K,Cout,Cin]: the input-gradient conv of that layer); weights and bias land in
// the context's arenas (valid until the next sr_conv2d_dev on the stream)
int conv_pack_weights_dev(sr_ctx* ctx, const float* d_hwio, const float* d_bias, int KS, int Cin, int Cout, int rot, ConvWeights* out, hipStream_t st);
int conv_prepack_dev(sr_ctx* ctx, const sr_pack_desc* descs, int n, hipStream_t st);
// rows_head = 1 (bf16 3x3 only): a conv whose Cin fits one 16-byte slice (an RGB head) is packed for the row-sliding kernel on one
// zero-padded 32-channel chunk instead of the thin kernel -- the caller then provides a 32-channel input view
int conv_pack_weights(sr_ctx* ctx, const float* hwio, const float* bias, int KS, int Cin, int Cout,
                      int dtype, ConvWeights* out, int rows_head = 0);
void conv_free_weights(sr_ctx* ctx, ConvWeights* w);
// x view must expose >= w.CinP channels starting at coff (extra ones multiplied by zero weights).
int conv_launch(sr_ctx* ctx, const ConvWeights& w, TensorView x, int B, int H, int W,
                void* y, int64_t y_cs, int y_coff, const ConvEpilogue& ep, hipStream_t st);
int conv_launch(sr_ctx* ctx, const ConvWeights& w, TensorView x, int B, int H, int W, TensorView y, const ConvEpilogue& ep, hipStream_t st);

// fused 1x1 projection: pack the HWIO kernel [1,1,64,16 * nblk] (+ bias) of the 1x1 conv that follows a 64-channel conv_rows conv
int proj_pack_weights(sr_ctx* ctx, const float* w_hwio, const float* bias, int cout, ProjWeights* out);
int pw2_pack_weights(sr_ctx* ctx, const float* w_hwio, const float* bias, int cin, int cout, int act, Pw2Weights* out);
void pw2_free_weights(sr_ctx* ctx, Pw2Weights* w);
void proj_free_weights(sr_ctx* ctx, ProjWeights* w);
// fused RGB tail: pack the second conv's HWIO kernel [3,3,64,c2] (+ bias), size of the partial-sum buffer, and the pass that adds the
// partial sums of the tiles covering an output pixel, applies bias / activation / alpha / clip and stores NHWC (bf16, or fp32 with out_f32)
int rgbtail_pack_weights(sr_ctx* ctx, const float* w2_hwio, const float* bias2, int c2, RgbTailWeights* out);
void rgbtail_free_weights(sr_ctx* ctx, RgbTailWeights* w);
int64_t rgbtail_partial_bytes(int B, int H, int W);
int rgbtail_finish_launch(sr_ctx* ctx, const RgbTailWeights& w, const float* part, int B, int H, int W, int act, float alpha, int clip01, void* y,
                          int64_t y_cs, int y_coff, int out_f32, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// fused pair of dense-block convs (dense_fused.hip)
// ---------------------------------------------------------------------------------------------
struct ChainWeights {
    void* w = nullptr; float* bias = nullptr;
    int ext = 0, nb0 = 0, nb1 = 0;   // external 32-channel chunks both convs read; 16-cout blocks of the first / second conv
    size_t bytes = 0;
};
int chain_pack_weights(sr_ctx* ctx, const float* wa_hwio, const float* ba, const float* wb_hwio, const float* bb, int ext, int nb0, int nb1, ChainWeights* out);
void chain_free_weights(sr_ctx* ctx, ChainWeights* w);
bool chain_supported(const ChainWeights& w, const TensorView& in, int W);
// conv1 of a dense block (64 -> 32, writes chunk 2 of the row-blocked buffer it reads) as a streaming line-buffer kernel (dense_fused.hip)
bool conv1_stream_supported(const ConvWeights& w, const TensorView& in, int W);
int conv1_stream_launch(sr_ctx* ctx, const ConvWeights& w, TensorView in, int B, int H, int W, hipStream_t st, bool seam = false);
int pack_pairs_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, hipStream_t st);
int unpack_pairs_launch(sr_ctx* ctx, const void* src, int64_t src_C, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, hipStream_t st);
int cell_pack_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, const CellGrid& g, hipStream_t st);
int cell_unpack_launch(sr_ctx* ctx, const void* src, int64_t src_C, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C, int dst_coff, const CellGrid& g, hipStream_t st);
int chain_launch(sr_ctx* ctx, const ChainWeights& w, TensorView in, int B, int H, int W, TensorView out, TensorView skip_o, float alpha, float beta_x,
                 float beta_o, hipStream_t st, bool seam = false);

// ---------------------------------------------------------------------------------------------
// attention core: o = softmax(q k^T) v per image, tokens N=H*W.
// qkv: [B,N,cs] with k(f) at channel koff (8 ch), q(g) at qoff (8 ch), v(h) at voff (32 ch);
// o: [B,N,o_cs] 32 channels at o_coff.  dtype = compute dtype of both.
// ---------------------------------------------------------------------------------------------
int attention_launch(sr_ctx* ctx, int dtype, const void* qkv, int64_t cs, int qoff, int koff, int voff,
                     int B, int N, void* o, int64_t o_cs, int o_coff, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// small ops (imgops.hip)
// ---------------------------------------------------------------------------------------------
// [B,H,W,C] of in_dtype -> [B,H,W,Cp] of out_dtype, channels >= C zero-filled, v*mul+add on real ones.
int nhwc_to_blocked_launch(sr_ctx* ctx, const void* src, int64_t src_cs, int src_coff, int B, int H, int W, int C, void* dst, int64_t dst_C,
                           int dst_coff, hipStream_t st);
int clock_probe_launch(sr_ctx* ctx, float* mhz_out, hipStream_t st);
int tap_copy_launch(sr_ctx* ctx, const void* src, int dtype, int blk, int64_t cs, int coff, int B, int H, int W, int C, float* dst, hipStream_t st, bool pairs = false);
int convert_pad_launch(sr_ctx* ctx, const void* x, int in_dtype, int64_t npix, int C, void* y, int out_dtype,
                       int Cp, float mul, float add, hipStream_t st);
int maxpool2_launch(sr_ctx* ctx, int dtype, const void* x, int B, int H, int W, int C, int64_t x_cs, void* y,
                    int64_t y_cs, hipStream_t st, CellGrid in_grid = CellGrid{}, CellGrid out_grid = CellGrid{});
int gap_launch(sr_ctx* ctx, int dtype, const void* x, int B, int HW, int C, int64_t x_cs, float* y, hipStream_t st);
// y[b,o] = act(sum_i x[b,i] w[i,o] + bias[o]); act: SR_ACT_LINEAR / _RELU / _LRELU, 100 = softmax over o, 101 = sigmoid.  fp32 in/out.
int dense_launch(sr_ctx* ctx, const float* x, const float* w, const float* bias, int B, int In, int Out, int act,
                 float* y, int y_dtype, void* y_typed, hipStream_t st);
int bicubic_launch(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW,
                   void* y, int out_dtype, int64_t y_cs, hipStream_t st);
int subsample2_launch(sr_ctx* ctx, int dtype, const void* x, int B, int H, int W, int C, int64_t x_cs, void* y, int64_t y_cs, hipStream_t st);
int vgg_preproc_launch(sr_ctx* ctx, const void* x, int in_dtype, int64_t npix, void* y, int out_dtype, int Cp, hipStream_t st);
int l1_launch(sr_ctx* ctx, const float* a, const float* b, int64_t n, float* out, hipStream_t st);
int spectral_l1_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float* out, hipStream_t st);
int wgrad_launch(sr_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int KS, float* dw, float* db, hipStream_t st);
int wgrad_launch_views(sr_ctx* ctx, const float* x, int64_t x_cs, const float* dy, int64_t dy_cs, int B, int H, int W, int Cin, int Cout, int KS, float* dw, float* db,
                       hipStream_t st);
int eltwise_views_launch(sr_ctx* ctx, int op, const float* a, int64_t a_cs, const float* b, int64_t b_cs, float alpha, float beta, float* out, int64_t o_cs, int64_t npix, int C,
                         hipStream_t st);
int adam_launch(sr_ctx* ctx, float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float omb1, float b2, float omb2, float eps,
                float gscale, hipStream_t st);
int eltwise_launch(sr_ctx* ctx, int op, const float* a, const float* b, float alpha, float beta, float* out, int64_t n, hipStream_t st);
int space_to_depth_launch(sr_ctx* ctx, const float* x, int B, int H, int W, int C, int r, float* y, hipStream_t st);
int matmul_launch(sr_ctx* ctx, const float* A, const float* B, float* C, int batch, int M, int N, int K, int tA, int tB, float alpha, hipStream_t st);
int softmax_rows_launch(sr_ctx* ctx, float* s, int64_t rows, int cols, hipStream_t st);
int softmax_bwd_launch(sr_ctx* ctx, const float* p, const float* dp, float* ds, int64_t rows, int cols, hipStream_t st);
int maxpool2_bwd_launch(sr_ctx* ctx, const float* x, const float* dy, int B, int H, int W, int C, float* dx, hipStream_t st);
int zero_insert2_launch(sr_ctx* ctx, const float* dy, int B, int H, int W, int C, float* out, hipStream_t st);
int spectral_l1_bwd_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float scale, float* da, hipStream_t st);
int resize_launch(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW, int interp, void* y, hipStream_t st);
int psnr_launch(sr_ctx* ctx, const float* a, const float* b, int B, int64_t n_per_image, float max_val, float* out,
                hipStream_t st);
int ssim_launch(sr_ctx* ctx, const float* a, const float* b, int B, int H, int W, int C, float max_val, float* out,
                hipStream_t st);
int mse_launch(sr_ctx* ctx, const float* a, const float* b, int64_t n, float* out, hipStream_t st);
int extract_patches_launch(sr_ctx* ctx, const float* img, int H, int W, int C, int patch, int stride, float mul,
                           float add, int out_dtype, void* out, int ny, int nx, hipStream_t st);
int overlap_add_launch(sr_ctx* ctx, const void* patches, int in_dtype, int H, int W, int C, int patch, int stride,
                       int scale, float mul, float add, int ny, int nx, float* out, hipStream_t st);

static inline int pad_amount(int n, int patch, int stride) {
    // reference: loading_methods.py:12-17
    int pad = (n % stride != 0) ? (patch - (n % stride)) % stride : 0;
    int extra = patch - stride;
    return pad > extra ? pad : extra;
}
