// api.hip -- the C ABI of libsr355.so (include/sr355.h) and the model graphs behind sr_forward.
//
// A model is a flat list of ops over numbered NHWC activation buffers, built once from the
// reference's hyper-parameters (SRCNN_model.py:45-53, EDSR_model.py:96-125, ESRGAN_model.py:303-345,
// VGG16_model.py:57-97).  All fusion decisions live here:
//   * bias/activation/residual-scale/skip adds/clip/depth_to_space ride in the conv epilogue;
//   * an ESRGAN dense block is a "virtual concat": one 64+4G-channel buffer, conv k reads channels
//     [0, 64+(k-1)G) and writes [64+(k-1)G, 64+kG); conv5 writes x + 0.2*conv into channels [0,64) of
//     the next block's buffer (three buffers rotate so the RRDB input survives for the outer skip);
//   * SelfAttention = one 1x1 conv producing f|g|h side by side, the streaming-softmax kernel, and
//     the 1x1 "v" conv with the residual add in its epilogue.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>

#include "common.h"

// =================================================================================================
// ctx
// =================================================================================================
void* sr_ctx::dalloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (alloc_cap > 0 && cur_bytes + (int64_t)bytes > alloc_cap) {
        err = "allocation of " + std::to_string(bytes) + " bytes exceeds the debug cap (sr_debug_set_alloc_cap)";
        return nullptr;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        return nullptr;
    }
    allocs[p] = bytes;
    cur_bytes += (int64_t)bytes;
    if (cur_bytes > peak_bytes) peak_bytes = cur_bytes;
    return p;
}
void sr_ctx::dfree(void* p) {
    if (!p) return;
    auto it = allocs.find(p);
    if (it != allocs.end()) { cur_bytes -= (int64_t)it->second; allocs.erase(it); }
    (void)hipFree(p);
}
// hipFuncAttributeMaxDynamicSharedMemorySize of `kernel` covers at least `bytes`.  The largest size set so far is remembered per kernel:
// kernels whose LDS grows with the image (the spectral-loss kernels: 56 / 80 bytes per pixel of width) raise it again when a wider
// image arrives (the first size used to be pinned and a later, wider launch failed with an opaque HIP error).
int sr_ctx::ensure_dyn_lds(const void* kernel, int bytes) {
    if (bytes > MAX_LDS_BYTES) return fail(SR_ERR_INVALID, "kernel needs " + std::to_string(bytes) + " bytes of LDS, a CU has " + std::to_string(MAX_LDS_BYTES));
    auto it = lds_attr_done.find(kernel);
    if (it != lds_attr_done.end() && it->second >= bytes) return SR_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(SR_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    lds_attr_done[kernel] = bytes;
    return SR_OK;
}
void* sr_ctx::scratch(size_t bytes) {
    if (bytes <= scratch_cap) return scratch_buf;
    if (scratch_buf) { (void)hipDeviceSynchronize(); dfree(scratch_buf); scratch_buf = nullptr; scratch_cap = 0; }
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    scratch_buf = dalloc(cap);
    if (scratch_buf) scratch_cap = cap;
    return scratch_buf;
}

int sr_ctx::cu_count() {
    if (num_cus <= 0) {
        hipDeviceProp_t prop;
        num_cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return num_cus;
}

void* sr_ctx::arena(Arena& a, size_t bytes, hipStream_t st) {
    if (bytes <= a.cap) return a.p;
    if (a.p) { (void)hipStreamSynchronize(st); dfree(a.p); a.p = nullptr; a.cap = 0; }
    const size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 4;
    a.p = dalloc(cap);
    if (a.p) a.cap = cap;
    return a.p;
}

int sr_ctx::prof_open(const std::string& name, double flops, double bytes, hipStream_t st) {
    if (!prof) return -1;
    int ni = -1;
    for (size_t i = 0; i < prof_names.size(); ++i) if (prof_names[i] == name) { ni = (int)i; break; }
    if (ni < 0) { prof_names.push_back(name); ni = (int)prof_names.size() - 1; }
    hipEvent_t e[2];
    for (int k = 0; k < 2; ++k) {
        if (!ev_pool.empty()) { e[k] = ev_pool.back(); ev_pool.pop_back(); }
        else if (hipEventCreate(&e[k]) != hipSuccess) return -1;
    }
    (void)hipEventRecord(e[0], st);
    prof_recs.push_back(ProfRec{ni, e[0], e[1], flops, bytes});
    return (int)prof_recs.size() - 1;
}
void sr_ctx::prof_close(int rec, hipStream_t st) {
    if (rec >= 0) (void)hipEventRecord(prof_recs[rec].e1, st);
}

// =================================================================================================
// model graph
// =================================================================================================
namespace {

struct Param {
    std::string name; int which; std::vector<int64_t> shape; std::vector<float> host; bool set = false;
    int64_t count() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};
struct Ref { int buf = -1; int coff = 0; };        // buf: >=0 internal, -1 none, -2 caller's y
enum OpKind { OP_CONVERT, OP_CONV, OP_ATTN, OP_POOL, OP_GAP, OP_DENSE, OP_TOBLK, OP_SUBSAMPLE, OP_PREPROC };
struct Op {
    OpKind kind = OP_CONV;
    Ref in, out, skip1, skip2;
    int conv = -1;
    int act = SR_ACT_LINEAR; float alpha = 1.f, beta1 = 0.f, beta2 = 0.f; int clip = 0, d2s = 1;
    int dw = -1, db = -1, In = 0, Out = 0;          // dense: param indices
    int chain = -1, chain_pos = 0;                  // conv: member (first / second) of m->chains[chain]
    int rgbtail = -1;                               // conv: first op of m->rgbtails[rgbtail] (the next op is the conv folded into this one's epilogue)
    int pack_alt = -1;                              // conv (ESRGAN trunk_conv): a free row-blocked buffer the two-up packed trunk output is unpacked into (sr_forward, 24-pixel-wide patches)
    int pw2 = -1;                                   // conv: the next op is the 1x1 conv m->pw2s[pw2], which this fp32 thin conv's epilogue can compute from its accumulators
    int proj = -1;                                  // conv: the next op is the 1x1 projection m->projs[proj], which this conv's epilogue can compute
};
struct BufSpec { int C = 0; int mul = 1; int shift = 0; bool vec = false; int Cbuf = 0; int blk = 0; int cshift = 0; int cells = 0; int cat = 0; };   // vec: fp32 [B,C]; blk: row-blocked (conv_common.h); shift: floor halvings (pooling), cshift: ceil halvings (stride-2 SAME convs); cells: small images may be packed in a CellGrid (common.h)
struct ConvPart { std::string name; int cout; float scale = 1.f; };   // scale: applied to kernel and bias when the conv is packed
struct ConvSpec { std::vector<ConvPart> parts; int KS = 3, Cin = 0, Cout = 0; ConvWeights w; int rows_head = 0; };   // rows_head: conv_pack_weights
// two consecutive convs of a dense block that run as ONE kernel when the shape allows (dense_fused.hip): ops[first], ops[first + 1]
struct ChainSpec { int conv_a = -1, conv_b = -1; int tail = 0; ChainWeights w; };
// a 64-cout 3x3 conv followed by the 3x3 conv to the image's <= 3 channels: one kernel + a finishing pass when the shape allows (conv_rows.hip)
struct RgbTailSpec { int conv_b = -1; RgbTailWeights w; };
// the f / g / h projections of a SelfAttention layer, computed in the epilogue of the conv that produces the layer's input (conv_rows.hip)
struct ProjSpec { int conv = -1; ProjWeights w; };
struct Pw2Spec { int conv = -1; int act = 0; Pw2Weights w; };

}  // namespace

struct sr_model {
    sr_ctx* ctx = nullptr;
    int kind = 0;
    sr_model_cfg cfg{};
    int T = SR_DTYPE_BF16;
    std::vector<Param> params;
    std::vector<ConvSpec> convs;
    std::vector<ChainSpec> chains;
    std::vector<RgbTailSpec> rgbtails;
    std::vector<ProjSpec> projs;
    std::vector<Pw2Spec> pw2s;
    std::vector<BufSpec> bufs;
    std::vector<void*> bufp;
    std::vector<float*> dense_dev;    // per param index (dense kernels / biases on device), else nullptr
    std::vector<Op> ops;
    int in_C = 3, out_C = 3, out_mul = 1, out_shift = 0; bool out_vec = false;
    bool finalized = false;
    std::vector<size_t> bufcap;       // bytes currently allocated per workspace buffer (grow-only)
    std::vector<CellGrid> grid_now;   // per workspace buffer: the packed layout its contents have now (gx = 0: plain NHWC)
    CellGrid cat_grid;                // ESRGAN generator: the cell grid the dense blocks' concat buffers hold now (sr_forward `cellpack`; gx = 0: none / unknown)
    int cat_dirty = -1;               // ... except this one, which trunk_conv's unpacked input was written to
    struct Tap { float* dst; int64_t cap; };
    std::unordered_map<int, Tap> taps; // diagnostic: op index -> device fp32 destination (sr_model_set_tap)
    std::vector<std::string> op_names; // per op, for sr_model_op_info

    int find_param(const std::string& n, int which) const {
        for (size_t i = 0; i < params.size(); ++i) if (params[i].which == which && params[i].name == n) return (int)i;
        return -1;
    }
    void free_bufs() { for (auto& p : bufp) { if (p) ctx->dfree(p); p = nullptr; } bufcap.assign(bufcap.size(), 0); grid_now.assign(grid_now.size(), CellGrid{}); cat_grid = CellGrid{}; cat_dirty = -1; }
};

namespace {

struct Builder {
    sr_model* m;
    int esz() const { return dtype_size(m->T); }
    int E() const { return 16 / esz(); }
    int buf(int C, int mul = 1, int shift = 0) {
        BufSpec b; b.C = C; b.mul = mul; b.shift = shift; b.Cbuf = round_up(C, 32);
        m->bufs.push_back(b); return (int)m->bufs.size() - 1;
    }
    int vecbuf(int C) { BufSpec b; b.C = C; b.vec = true; b.Cbuf = C; m->bufs.push_back(b); return (int)m->bufs.size() - 1; }
    void add_layer_params(const std::string& name, std::vector<int64_t> kshape) {
        Param k; k.name = name; k.which = SR_WEIGHT_KERNEL; k.shape = kshape; m->params.push_back(k);
        Param b; b.name = name; b.which = SR_WEIGHT_BIAS; b.shape = {kshape.back()}; m->params.push_back(b);
    }
    int conv_spec(std::vector<ConvPart> parts, int KS, int Cin) {
        ConvSpec c; c.parts = parts; c.KS = KS; c.Cin = Cin; c.Cout = 0;
        for (auto& p : parts) { c.Cout += p.cout; add_layer_params(p.name, {KS, KS, Cin, p.cout}); }
        m->convs.push_back(c); return (int)m->convs.size() - 1;
    }
    Op& conv(const std::string& name, int KS, int Cin, int Cout, Ref in, Ref out, int act = SR_ACT_LINEAR) {
        Op o; o.kind = OP_CONV; o.conv = conv_spec({{name, Cout}}, KS, Cin); o.in = in; o.out = out; o.act = act;
        m->ops.push_back(o); return m->ops.back();
    }
    // the same Keras layer executed a second time into another buffer (no new parameters)
    Op& conv_again(int spec, Ref in, Ref out, int act = SR_ACT_LINEAR) {
        Op o; o.kind = OP_CONV; o.conv = spec; o.in = in; o.out = out; o.act = act;
        m->ops.push_back(o); return m->ops.back();
    }
    // SelfAttention(64) on `x` (64 ch) -> `y` (64 ch); qkv: 64-ch scratch, ao: 32-ch scratch (same resolution)
    void self_attention(const std::string& name, int x, int y, int qkv, int ao) {
        // bf16: the key projection f is packed pre-multiplied by log2(e) -- the attention kernel works in the exp2 domain and has the
        // matrix core subtract the running max (attention.hip), so nothing is left to scale per score
        const float kscale = m->T == SR_DTYPE_BF16 ? 1.4426950408889634f : 1.f;
        Op p; p.kind = OP_CONV; p.conv = conv_spec({{name + "_f", 8, kscale}, {name + "_g", 8}, {name + "_h", 32}}, 1, 64);
        p.in = {x, 0}; p.out = {qkv, 0};
        if (m->T == SR_DTYPE_BF16 && !m->ops.empty() && m->ops.back().kind == OP_CONV && m->ops.back().out.buf == x && m->ops.back().out.coff == 0) {
            // the conv that has just produced x can compute these 48 channels in its epilogue (conv_rows.hip, rows_epilogue_proj); sr_forward decides per call
            ProjSpec ps; ps.conv = p.conv;
            m->projs.push_back(ps);
            m->ops.back().proj = (int)m->projs.size() - 1;
        }
        m->ops.push_back(p);
        Op a; a.kind = OP_ATTN; a.in = {qkv, 0}; a.out = {ao, 0}; m->ops.push_back(a);
        Op& v = conv(name + "_v", 1, 32, 64, {ao, 0}, {y, 0});
        v.skip1 = {x, 0}; v.beta1 = 1.f;
    }
};

int build_srcnn(sr_model* m) {
    Builder b{m};
    const int C = m->cfg.channels;
    m->in_C = C; m->out_C = C; m->out_mul = 1;
    const int x0 = b.buf(b.E()), c1 = b.buf(96), c2 = b.buf(32);
    m->bufs[x0].Cbuf = b.E();
    Op cv; cv.kind = OP_CONVERT; cv.out = {x0, 0}; m->ops.push_back(cv);
    b.conv("conv2d", 9, C, 96, {x0, 0}, {c1, 0}, SR_ACT_RELU);
    b.conv("conv2d_1", 1, 96, 32, {c1, 0}, {c2, 0}, SR_ACT_RELU);
    if (m->T == SR_DTYPE_F32 && C == 3) {
        // the 96-channel intermediate must never reach HBM -- the head's epilogue computes this 1x1 from its accumulators
        // (conv.hip, conv_thin_kernel PW2); sr_forward decides per call (a tap on conv2d, or sr_debug_set_fused without bit 8, runs the two convs)
        Pw2Spec ps; ps.conv = m->ops.back().conv; ps.act = SR_ACT_RELU;
        m->pw2s.push_back(ps);
        m->ops[m->ops.size() - 2].pw2 = (int)m->pw2s.size() - 1;
    }
    b.conv("conv2d_2", 5, 32, C, {c2, 0}, {-2, 0});
    return SR_OK;
}

int build_edsr(sr_model* m) {
    Builder b{m};
    const sr_model_cfg& c = m->cfg;
    const int C = c.channels, F = c.num_filters, s = c.scale_factor;
    if (s != 2 && s != 3 && s != 4) return m->ctx->fail(SR_ERR_INVALID, "Scale factor " + std::to_string(s) + " not supported. Use 2, 3, or 4.");
    m->in_C = C; m->out_C = C; m->out_mul = s;
    int li = 0;
    auto lname = [&]() { std::string n = li == 0 ? "conv2d" : "conv2d_" + std::to_string(li); ++li; return n; };
    const int x0 = b.buf(b.E());
    m->bufs[x0].Cbuf = b.E();
    const int head = b.buf(F), tmp = b.buf(F), ra = b.buf(F), rb = b.buf(F);
    Op cv; cv.kind = OP_CONVERT; cv.out = {x0, 0}; m->ops.push_back(cv);
    b.conv(lname(), 3, C, F, {x0, 0}, {head, 0});
    int cur = head;
    for (int i = 0; i < c.num_blocks; ++i) {
        const int nxt = (cur == ra) ? rb : ra;
        b.conv(lname(), 3, F, F, {cur, 0}, {tmp, 0}, SR_ACT_RELU);
        Op& o = b.conv(lname(), 3, F, F, {tmp, 0}, {nxt, 0});
        o.alpha = c.res_scaling; o.skip1 = {cur, 0}; o.beta1 = 1.f;      // x + res_scaling*conv (EDSR_model.py:68-72)
        cur = nxt;
    }
    const int body = (cur == ra) ? rb : ra;
    { Op& o = b.conv(lname(), 3, F, F, {cur, 0}, {body, 0}); o.skip1 = {head, 0}; o.beta1 = 1.f; }
    int up = body, mul = 1;
    const int nup = s == 4 ? 2 : 1, r = s == 4 ? 2 : s;
    for (int i = 0; i < nup; ++i) {
        mul *= r;
        const int ub = b.buf(F, mul);
        Op& o = b.conv(lname(), 3, F, F * r * r, {up, 0}, {ub, 0}); o.d2s = r;
        up = ub;
    }
    { Op& o = b.conv(lname(), 3, F, C, {up, 0}, {-2, 0}); o.clip = 1; }
    return SR_OK;
}

int build_esrgan(sr_model* m) {
    Builder b{m};
    const sr_model_cfg& c = m->cfg;
    const int C = c.channels, G = c.growth_channels, s = c.scale_factor;
    if (s < 1 || (s & (s - 1)) != 0) return m->ctx->fail(SR_ERR_INVALID, "ESRGAN scale_factor must be a power of two");
    if (G <= 0 || c.num_blocks < 0) return m->ctx->fail(SR_ERR_INVALID, "ESRGAN growth_channels/num_rrdb_blocks invalid");
    m->in_C = C; m->out_C = C; m->out_mul = s;
    const int CC = 64 + 4 * G;
    const int x0 = b.buf(b.E());
    // bf16: the RGB head on the row-sliding kernel over one zero-padded 32-channel chunk (as in build_vgg16: whole-line stores, 64 couts per workgroup)
    const bool head_rows = m->T == SR_DTYPE_BF16;
    m->bufs[x0].Cbuf = head_rows ? 32 : b.E();
    const int trunk = b.buf(64);
    int cat[3] = {b.buf(CC), b.buf(CC), b.buf(CC)};
    // The concat buffers are only ever touched by convs (written by initial_conv / the dense convs, read by the bf16 3x3
    // kernel): keep them row-blocked so that a 32-channel chunk of a tile row is one contiguous run of whole 128-byte lines.
    // (round 4: any growth width whose concat tensor is a whole number of 32-channel blocks -- the reference's notebook trains G = 8: 96 channels; a growth conv's
    //  output slice then starts inside a block, which the epilogue's per-lane addressing places)
    static const bool blk_g32_only = getenv("SR355_BLOCKED_G32_ONLY") != nullptr;       // A/B switch (diagnostic)
    if (m->T == SR_DTYPE_BF16 && (G % 32 == 0 || (G % 8 == 0 && !blk_g32_only)))
        for (int i = 0; i < 3; ++i) m->bufs[cat[i]].blk = 1;
    for (int i = 0; i < 3; ++i) m->bufs[cat[i]].cat = 1;
    Op cv; cv.kind = OP_CONVERT; cv.out = {x0, 0}; m->ops.push_back(cv);
    b.conv("initial_conv", 3, C, 64, {x0, 0}, {trunk, 0});
    if (head_rows) m->convs.back().rows_head = 1;
    if (m->bufs[cat[0]].blk) {        // the trunk input also goes into the first concat buffer: a layout-changing copy of 64 channels
        Op t; t.kind = OP_TOBLK; t.in = {trunk, 0}; t.out = {cat[0], 0}; m->ops.push_back(t);
    } else {
        b.conv_again((int)m->convs.size() - 1, {x0, 0}, {cat[0], 0});   // same layer again, straight into the first concat buffer
    }
    int X = 0, Y = 1, Z = 2;                                          // cat[X] holds the RRDB input in channels [0,64)
    for (int r = 0; r < c.num_blocks; ++r) {
        const int src[3] = {X, Y, Z}, dst[3] = {Y, Z, Y};
        for (int d = 0; d < 3; ++d) {
            const std::string dn = "rrdb_" + std::to_string(r) + "_dense" + std::to_string(d + 1);
            const int I = cat[src[d]], O = cat[dst[d]];
            const size_t first_op = m->ops.size();
            for (int k = 1; k <= 4; ++k)
                b.conv(dn + "_conv" + std::to_string(k), 3, 64 + (k - 1) * G, G, {I, 0}, {I, 64 + (k - 1) * G}, SR_ACT_RELU);
            Op& o = b.conv(dn + "_conv5", 3, CC, 64, {I, 0}, {O, 0});
            if (d < 2) { o.alpha = 0.2f; o.skip1 = {I, 0}; o.beta1 = 1.f; }                 // x + 0.2*conv5 (ESRGAN_model.py:249-252)
            else { o.alpha = 0.04f; o.skip1 = {cat[X], 0}; o.beta1 = 1.f; o.skip2 = {I, 0}; o.beta2 = 0.2f; }
            // d == 2: rrdb_in + 0.2*(x + 0.2*conv5)  (ESRGAN_model.py:277-280)
            if (m->bufs[I].blk && G == 32) {
                // row-blocked buffers, 32 growth channels: conv2+conv3 and conv4+conv5 can each run as one line-buffered kernel on
                // 48-pixel-wide images (dense_fused.hip); sr_forward decides per call and otherwise runs the ops one by one
                auto link = [&](size_t opa, int tail) {
                    ChainSpec cs; cs.conv_a = m->ops[opa].conv; cs.conv_b = m->ops[opa + 1].conv; cs.tail = tail;
                    m->chains.push_back(cs);
                    m->ops[opa].chain = m->ops[opa + 1].chain = (int)m->chains.size() - 1;
                    m->ops[opa].chain_pos = 0; m->ops[opa + 1].chain_pos = 1;
                };
                link(first_op + 1, 0);
                link(first_op + 3, 1);
            }
        }
        const int nX = Y, nY = Z, nZ = X; X = nX; Y = nY; Z = nZ;
    }
    const int t2 = b.buf(64);
    { Op& o = b.conv("trunk_conv", 3, 64, 64, {cat[X], 0}, {t2, 0}); o.skip1 = {trunk, 0}; o.beta1 = 1.f; o.pack_alt = cat[Y]; }
    int cur = t2;
    if (c.use_attention) {
        const int qkv = b.buf(64), ao = b.buf(32), t3 = b.buf(64);
        b.self_attention("self_attention_trunk", cur, t3, qkv, ao);
        cur = t3;
    }
    int mul = 1, nup = 0;
    for (int t = s; t > 1; t >>= 1) ++nup;
    for (int i = 0; i < nup; ++i) {
        mul *= 2;
        const int u = b.buf(64, mul);
        Op& o = b.conv("upsample_" + std::to_string(i) + "_conv", 3, 64, 256, {cur, 0}, {u, 0}, SR_ACT_LRELU);
        o.d2s = 2;   // LeakyReLU is elementwise: applying it before the shuffle is the same function
        cur = u;
        if (i == 0 && c.use_attention) {
            const int qkv = b.buf(64, mul), ao = b.buf(32, mul), ua = b.buf(64, mul);
            b.self_attention("self_attention_upsample_0", cur, ua, qkv, ao);
            cur = ua;
        }
    }
    const int f1 = b.buf(64, mul);
    b.conv("final_conv1", 3, 64, 64, {cur, 0}, {f1, 0}, SR_ACT_RELU);
    b.conv("final_conv2", 3, 64, C, {f1, 0}, {-2, 0}, SR_ACT_TANH);
    if (m->T == SR_DTYPE_BF16 && C <= 3) {
        // final_conv2 can ride in final_conv1's epilogue (conv_rows.hip, rows_fuse2): the 64-channel image at the output resolution is
        // then never written; sr_forward decides per call (a tap on final_conv1, or sr_debug_set_fused without bit 2, runs the two convs)
        RgbTailSpec rt; rt.conv_b = m->ops.back().conv;
        m->rgbtails.push_back(rt);
        m->ops[m->ops.size() - 2].rgbtail = (int)m->rgbtails.size() - 1;
    }
    return SR_OK;
}

int build_vgg16(sr_model* m) {
    Builder b{m};
    const int nc = m->cfg.num_classes;
    if (nc <= 0) return m->ctx->fail(SR_ERR_INVALID, "num_classes must be positive");
    m->in_C = 3; m->out_C = nc; m->out_vec = true;
    static const int cfg[5][2] = {{2, 64}, {2, 128}, {3, 256}, {3, 512}, {3, 512}};
    const int x0 = b.buf(b.E());
    // bf16: the RGB head runs on the row-sliding kernel over ONE zero-padded 32-channel chunk (64-cout workgroups, whole-line 16-byte stores)
    // instead of the thin kernel (32-cout workgroups, 8-byte stores through the generic epilogue): 0.88 -> 0.4 ms per 1024 patches of 96 x 96.
    // 29 of the chunk's 32 channels multiply zeros -- the layer is bound by its 64-channel output stream either way.
    const bool head_rows = m->T == SR_DTYPE_BF16;
    m->bufs[x0].Cbuf = head_rows ? 32 : b.E();
    Op cv; cv.kind = OP_CONVERT; cv.out = {x0, 0}; m->ops.push_back(cv);
    int cur = x0, cin = 3;
    for (int blk = 0; blk < 5; ++blk) {
        for (int k = 0; k < cfg[blk][0]; ++k) {
            const int o = b.buf(cfg[blk][1], 1, blk);
            b.conv("block" + std::to_string(blk + 1) + "_conv" + std::to_string(k + 1), 3, cin, cfg[blk][1], {cur, 0}, {o, 0}, SR_ACT_RELU);
            if (blk == 0 && k == 0 && head_rows) m->convs.back().rows_head = 1;
            // block 5 works on (patch / 16)^2 images -- 6 x 6 for the reference's 96-pixel patches: its buffers may hold the batch packed in a
            // CellGrid (sr_forward decides per call; the pool in front writes that layout, the pool behind reads it)
            if (blk == 4 && m->T == SR_DTYPE_BF16) m->bufs[o].cells = 1;
            cur = o; cin = cfg[blk][1];
        }
        const int pb = b.buf(cin, 1, blk + 1);
        if (blk == 3 && m->T == SR_DTYPE_BF16) m->bufs[pb].cells = 1;
        Op p; p.kind = OP_POOL; p.in = {cur, 0}; p.out = {pb, 0}; m->ops.push_back(p);
        cur = pb;
    }
    const int g = b.vecbuf(512), d1 = b.vecbuf(256);
    { Op o; o.kind = OP_GAP; o.in = {cur, 0}; o.out = {g, 0}; m->ops.push_back(o); }
    b.add_layer_params("dense", {512, 256});
    { Op o; o.kind = OP_DENSE; o.in = {g, 0}; o.out = {d1, 0}; o.In = 512; o.Out = 256; o.act = SR_ACT_RELU;
      o.dw = m->find_param("dense", SR_WEIGHT_KERNEL); o.db = m->find_param("dense", SR_WEIGHT_BIAS); m->ops.push_back(o); }
    b.add_layer_params("predictions", {256, nc});
    { Op o; o.kind = OP_DENSE; o.in = {d1, 0}; o.out = {-2, 0}; o.In = 256; o.Out = nc; o.act = 100;
      o.dw = m->find_param("predictions", SR_WEIGHT_KERNEL); o.db = m->find_param("predictions", SR_WEIGHT_BIAS); m->ops.push_back(o); }
    return SR_OK;
}

// ESRGAN discriminator, inference graph (ESRGAN_model.py:347-377 with training=False, as ESRGAN.evaluate runs it :810-812): six 3x3
// SAME convs with strides 1,2,1,2,1,2 + LeakyReLU(0.2), GAP, Dense 256 + LeakyReLU, Dense 1 sigmoid.  The SpectralNormalization
// wrappers only act when training (they renormalise the stored kernel in place): at inference the stored kernel is the
// layer.  A stride-2 SAME conv = the stride-1 conv (LeakyReLU fused: elementwise) + a pick of every second position (imgops.hip).
int build_discriminator(sr_model* m) {
    Builder b{m};
    m->in_C = 3; m->out_C = 1; m->out_vec = true;
    const int x0 = b.buf(b.E());
    m->bufs[x0].Cbuf = b.E();
    Op cv; cv.kind = OP_CONVERT; cv.out = {x0, 0}; m->ops.push_back(cv);
    static const int filt[6] = {64, 64, 64, 128, 128, 256}, strd[6] = {1, 2, 1, 2, 1, 2};
    int cur = x0, cin = 3, halv = 0;
    for (int i = 0; i < 6; ++i) {
        const int o = b.buf(filt[i]);
        m->bufs[o].cshift = halv;
        b.conv("disc_conv" + std::to_string(i + 1), 3, cin, filt[i], {cur, 0}, {o, 0}, SR_ACT_LRELU);
        cur = o; cin = filt[i];
        if (strd[i] == 2) {
            ++halv;
            const int s2 = b.buf(filt[i]);
            m->bufs[s2].cshift = halv;
            Op sp; sp.kind = OP_SUBSAMPLE; sp.in = {cur, 0}; sp.out = {s2, 0}; m->ops.push_back(sp);
            cur = s2;
        }
    }
    const int g = b.vecbuf(256), d1 = b.vecbuf(256);
    { Op o; o.kind = OP_GAP; o.in = {cur, 0}; o.out = {g, 0}; m->ops.push_back(o); }
    b.add_layer_params("disc_dense1", {256, 256});
    { Op o; o.kind = OP_DENSE; o.in = {g, 0}; o.out = {d1, 0}; o.In = 256; o.Out = 256; o.act = SR_ACT_LRELU;
      o.dw = m->find_param("disc_dense1", SR_WEIGHT_KERNEL); o.db = m->find_param("disc_dense1", SR_WEIGHT_BIAS); m->ops.push_back(o); }
    b.add_layer_params("disc_output", {256, 1});
    { Op o; o.kind = OP_DENSE; o.in = {d1, 0}; o.out = {-2, 0}; o.In = 256; o.Out = 1; o.act = 101;
      o.dw = m->find_param("disc_output", SR_WEIGHT_KERNEL); o.db = m->find_param("disc_output", SR_WEIGHT_BIAS); m->ops.push_back(o); }
    return SR_OK;
}

// VGG19 perceptual-feature extractor (ESRGAN_model.py:379-408): caffe-mode preprocessing of a [-1,1] image, then keras VGG19 without
// top up to block5_conv4 (ReLU included) -> [B, H/16, W/16, 512].
int build_vgg19_features(sr_model* m) {
    Builder b{m};
    m->in_C = 3; m->out_C = 512; m->out_mul = 1; m->out_shift = 4;
    static const int cfg[5][2] = {{2, 64}, {2, 128}, {4, 256}, {4, 512}, {4, 512}};
    const int x0 = b.buf(b.E());
    m->bufs[x0].Cbuf = b.E();
    Op pp; pp.kind = OP_PREPROC; pp.out = {x0, 0}; m->ops.push_back(pp);
    int cur = x0, cin = 3;
    for (int blk = 0; blk < 5; ++blk) {
        for (int k = 0; k < cfg[blk][0]; ++k) {
            const bool last = blk == 4 && k == cfg[blk][0] - 1;
            const int o = last ? -2 : b.buf(cfg[blk][1], 1, blk);
            b.conv("block" + std::to_string(blk + 1) + "_conv" + std::to_string(k + 1), 3, cin, cfg[blk][1], {cur, 0}, {o, 0}, SR_ACT_RELU);
            cur = o; cin = cfg[blk][1];
        }
        if (blk == 4) break;
        const int pb = b.buf(cin, 1, blk + 1);
        Op p; p.kind = OP_POOL; p.in = {cur, 0}; p.out = {pb, 0}; m->ops.push_back(p);
        cur = pb;
    }
    return SR_OK;
}

inline void buf_hw(const BufSpec& b, int H, int W, int* h, int* w) {
    *h = (H * b.mul) >> b.shift; *w = (W * b.mul) >> b.shift;
    for (int i = 0; i < b.cshift; ++i) { *h = (*h + 1) / 2; *w = (*w + 1) / 2; }
}

// Workspaces grow on demand and are never shrunk.  Every buffer is [pixels][Cbuf]: the position of the pad
// channels inside a pixel does not depend on (B,H,W), so zeroing them once at allocation stays valid for every
// later shape (no kernel ever writes a pad channel).
// On an allocation failure every workspace buffer is released (after a device sync: earlier forwards may still be running on
// them) so that a retry with a smaller batch starts from a clean slate instead of from the half-grown oversized set.
// The packed layout of B images of h x w pixels (CellGrid, common.h), or gx = 0 where packing does not pay: the 64-cout kernel issues MFMAs for
// 12 x 16 output tiles, so a plain image uses h w / (ceil(h / 12) 12 ceil(w / 16) 16) of them and the grid (h / (h + 1)) (gx w / 16).
CellGrid cell_grid_for(int B, int h, int w, int Cbuf) {
    CellGrid g;
    if (h < 1 || w < 1 || w + 1 > 16) return g;
    const int gx = 16 / (w + 1);
    const double plain = (double)h * w / ((double)((h + 11) / 12 * 12) * ((w + 15) / 16 * 16));
    const double packed = (double)h / (h + 1) * gx * w / 16.0;
    if (packed < 1.3 * plain) return g;
    const int64_t Hv = (int64_t)((B + gx - 1) / gx) * (h + 1), Wv = (int64_t)gx * (w + 1);
    if (Hv * Wv * Cbuf >= ((int64_t)1 << 31)) return g;           // the conv kernels address an image with 32-bit offsets
    g.gx = gx; g.ch = h + 1; g.cw = w + 1; g.Hv = (int)Hv; g.Wv = (int)Wv;
    return g;
}

// use_cells: buffers marked `cells` adopt the packed layout for this forward where it pays (never with taps set: the tap copy reads plain NHWC)
int ensure_workspace(sr_model* m, int B, int H, int W, hipStream_t st, bool use_cells) {
    if (m->bufp.size() != m->bufs.size()) { m->bufp.assign(m->bufs.size(), nullptr); m->bufcap.assign(m->bufs.size(), 0); m->grid_now.assign(m->bufs.size(), CellGrid{}); }
    for (size_t i = 0; i < m->bufs.size(); ++i) {
        const BufSpec& b = m->bufs[i];
        size_t bytes;
        CellGrid want;
        if (b.vec) bytes = (size_t)B * b.C * sizeof(float);
        else {
            int h, w; buf_hw(b, H, W, &h, &w);
            bytes = (size_t)B * h * w * b.Cbuf * dtype_size(m->T) + 4096;
            if (b.cells && use_cells) {
                want = cell_grid_for(B, h, w, b.Cbuf);
                if (want.gx) bytes = std::max(bytes, (size_t)want.Hv * want.Wv * b.Cbuf * dtype_size(m->T) + 4096);
            }
        }
        if (bytes <= m->bufcap[i]) {
            // the separators of a packed layout must be zeros: adopting one over other contents (a plain forward, another grid) clears the buffer
            const CellGrid& have = m->grid_now[i];                  // (the cells of another batch size lie in the same places: nothing to clear)
            if (want.gx && !(have.gx == want.gx && have.ch == want.ch && have.cw == want.cw && have.Wv == want.Wv)) SR_HIP(m->ctx, hipMemsetAsync(m->bufp[i], 0, m->bufcap[i], st));
            m->grid_now[i] = want;
            continue;
        }
        m->grid_now[i] = want;                                      // (a new allocation is zeroed below)
        if (m->bufp[i]) { SR_HIP(m->ctx, hipDeviceSynchronize()); m->ctx->dfree(m->bufp[i]); m->bufp[i] = nullptr; m->bufcap[i] = 0; }
        m->bufp[i] = m->ctx->dalloc(bytes);
        if (!m->bufp[i]) {
            const std::string why = m->ctx->err;
            (void)hipDeviceSynchronize();
            m->free_bufs();
            return m->ctx->fail(SR_ERR_OOM, "workspace for [" + std::to_string(B) + "," + std::to_string(H) + "," + std::to_string(W) + "]: " + why +
                                                " (all workspaces of this model were released)");
        }
        SR_HIP(m->ctx, hipMemsetAsync(m->bufp[i], 0, bytes, st));   // on the forward's stream: ordered before its first kernel
        m->bufcap[i] = bytes;
    }
    return SR_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int sr_init(int device_id, sr_ctx** out) {
    if (!out) return SR_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return SR_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return SR_ERR_HIP;
    sr_ctx* c = new sr_ctx();
    c->device = device_id;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { delete c; return SR_ERR_HIP; }
    c->zero_page = c->dalloc(sr_ctx::ZERO_PAGE_BYTES);
    if (!c->zero_page) { sr_destroy(c); return SR_ERR_OOM; }
    if (hipMemset(c->zero_page, 0, sr_ctx::ZERO_PAGE_BYTES) != hipSuccess) { sr_destroy(c); return SR_ERR_HIP; }
    *out = c;
    return SR_OK;
}

void sr_destroy(sr_ctx* ctx) {
    DeviceGuard dg_(ctx);
    if (!ctx) return;
    (void)hipDeviceSynchronize();
    std::vector<void*> ps;
    for (auto& kv : ctx->allocs) ps.push_back(kv.first);
    for (void* p : ps) ctx->dfree(p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto& e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
}

const char* sr_last_error(sr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int sr_mem_info(sr_ctx* ctx, int64_t* current_bytes, int64_t* peak_bytes) {
    if (!ctx) return SR_ERR_INVALID;
    if (current_bytes) *current_bytes = ctx->cur_bytes;
    if (peak_bytes) *peak_bytes = ctx->peak_bytes;
    return SR_OK;
}

int sr_last_forward_ms(sr_ctx* ctx, float* ms) {
    DeviceGuard dg_(ctx);
    if (!ctx || !ms) return SR_ERR_INVALID;
    if (!ctx->timed) return ctx->fail(SR_ERR_STATE, "no forward has run on this ctx");
    SR_HIP(ctx, hipEventSynchronize(ctx->ev1));
    SR_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return SR_OK;
}

int64_t sr_debug_stamp_bytes_needed(int which, int64_t workgroups) {
    if (which == 0) return workgroups < 0 ? -1 : workgroups * 16 * (int64_t)sizeof(unsigned long long);
    if (which == 1) return (int64_t)64 * 4 * 64 * 4 * (int64_t)sizeof(unsigned long long);
    return -1;
}

int sr_debug_set_stamp_buffer(sr_ctx* ctx, void* device_u64_buffer, int64_t capacity_bytes) {
    if (!ctx) return SR_ERR_INVALID;
    if (device_u64_buffer && capacity_bytes < sr_debug_stamp_bytes_needed(0, 1))
        return ctx->fail(SR_ERR_INVALID, "sr_debug_set_stamp_buffer: the buffer does not hold one workgroup's stamps");
    ctx->stamp_buf = static_cast<unsigned long long*>(device_u64_buffer);
    ctx->stamp_cap = device_u64_buffer ? capacity_bytes : 0;
    return SR_OK;
}

int sr_measure_clock(sr_ctx* ctx, float* mhz, void* stream) {
    if (!ctx || !mhz) return SR_ERR_INVALID;
    DeviceGuard dg_(ctx);
    return clock_probe_launch(ctx, mhz, static_cast<hipStream_t>(stream));
}

int sr_debug_set_chain_stamp_buffer(sr_ctx* ctx, void* device_u64_buffer, int64_t capacity_bytes) {
    if (!ctx) return SR_ERR_INVALID;
    if (device_u64_buffer && capacity_bytes < sr_debug_stamp_bytes_needed(1, 0))
        return ctx->fail(SR_ERR_INVALID, "sr_debug_set_chain_stamp_buffer: the buffer is smaller than the stamped kernels write (sr_debug_stamp_bytes_needed(1, 0))");
    ctx->chain_stamp_buf = static_cast<unsigned long long*>(device_u64_buffer);
    const char* skip = getenv("SR355_CHAIN_STAMP_SKIP");
    ctx->chain_stamp_skip = skip ? atoi(skip) : -1;
    return SR_OK;
}

int sr_debug_set_fused(sr_ctx* ctx, int mask, int max_workgroups) {
    if (!ctx) return SR_ERR_INVALID;
    ctx->chain_mask = mask;
    ctx->chain_max_wgs = max_workgroups > 0 ? max_workgroups : 0;
    return SR_OK;
}

int sr_debug_set_alloc_cap(sr_ctx* ctx, int64_t bytes) {
    if (!ctx) return SR_ERR_INVALID;
    ctx->alloc_cap = bytes > 0 ? bytes : 0;
    return SR_OK;
}

int sr_profile_begin(sr_ctx* ctx) {
    if (!ctx) return SR_ERR_INVALID;
    for (auto& r : ctx->prof_recs) { ctx->ev_pool.push_back(r.e0); ctx->ev_pool.push_back(r.e1); }
    ctx->prof_recs.clear();
    ctx->prof = true;
    return SR_OK;
}

int sr_profile_end(sr_ctx* ctx, char* json, int64_t cap) {
    DeviceGuard dg_(ctx);
    if (!ctx || !json || cap <= 0) return SR_ERR_INVALID;
    ctx->prof = false;
    SR_HIP(ctx, hipDeviceSynchronize());
    struct Agg { int64_t n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<Agg> agg(ctx->prof_names.size());
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg& a = agg[r.name];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    std::string s = "[";
    bool first = true;
    for (size_t i = 0; i < agg.size(); ++i) {
        if (!agg[i].n) continue;
        char buf[512];
        snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"launches\": %lld, \"total_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", ctx->prof_names[i].c_str(), (long long)agg[i].n, agg[i].ms, agg[i].flops, agg[i].bytes);
        s += buf;
        first = false;
    }
    s += "]";
    for (auto& r : ctx->prof_recs) { ctx->ev_pool.push_back(r.e0); ctx->ev_pool.push_back(r.e1); }
    ctx->prof_recs.clear();
    if ((int64_t)s.size() + 1 > cap) return ctx->fail(SR_ERR_CAPACITY, "profile buffer too small");
    memcpy(json, s.c_str(), s.size() + 1);
    return SR_OK;
}

int sr_model_create(sr_ctx* ctx, int kind, const sr_model_cfg* cfg, sr_model** out) {
    if (!ctx || !cfg || !out) return SR_ERR_INVALID;
    *out = nullptr;
    if (cfg->compute_dtype != SR_DTYPE_F32 && cfg->compute_dtype != SR_DTYPE_BF16)
        return ctx->fail(SR_ERR_INVALID, "compute_dtype must be SR_DTYPE_F32 or SR_DTYPE_BF16");
    std::unique_ptr<sr_model> m(new sr_model());
    m->ctx = ctx; m->kind = kind; m->cfg = *cfg; m->T = cfg->compute_dtype;
    if (m->cfg.channels <= 0) m->cfg.channels = 3;
    if (m->cfg.channels > 16 / dtype_size(m->T))
        return ctx->fail(SR_ERR_INVALID, "channels must fit one 16-byte slice (<= 4 for f32, <= 8 for bf16)");
    int rc;
    switch (kind) {
        case SR_MODEL_SRCNN: rc = build_srcnn(m.get()); break;
        case SR_MODEL_EDSR: rc = build_edsr(m.get()); break;
        case SR_MODEL_ESRGAN_G: rc = build_esrgan(m.get()); break;
        case SR_MODEL_VGG16: rc = build_vgg16(m.get()); break;
        case SR_MODEL_ESRGAN_D: rc = build_discriminator(m.get()); break;
        case SR_MODEL_VGG19_FEATURES: rc = build_vgg19_features(m.get()); break;
        default: return ctx->fail(SR_ERR_INVALID, "unknown model kind");
    }
    if (rc) return rc;
    m->dense_dev.assign(m->params.size(), nullptr);
    *out = m.release();
    return SR_OK;
}

void sr_model_destroy(sr_model* m) {
    DeviceGuard dg_(m ? m->ctx : nullptr);
    if (!m) return;
    (void)hipDeviceSynchronize();
    m->free_bufs();
    for (auto& c : m->convs) conv_free_weights(m->ctx, &c.w);
    for (auto& ch : m->chains) chain_free_weights(m->ctx, &ch.w);
    for (auto& rt : m->rgbtails) rgbtail_free_weights(m->ctx, &rt.w);
    for (auto& pj : m->projs) proj_free_weights(m->ctx, &pj.w);
    for (auto& pw : m->pw2s) pw2_free_weights(m->ctx, &pw.w);
    for (auto& p : m->dense_dev) if (p) m->ctx->dfree(p);
    delete m;
}

int sr_model_release_workspace(sr_model* m) {
    if (!m) return SR_ERR_INVALID;
    DeviceGuard dg_(m->ctx);
    SR_HIP(m->ctx, hipDeviceSynchronize());
    m->free_bufs();
    return SR_OK;
}

// ---- diagnostics: the op list and per-op output taps (stage-by-stage parity traces in tests/) ------------------------
static void op_out_view(const sr_model* m, const Op& op, int* C, int* mul, int* shift, int* cshift = nullptr) {
    *C = 0; *mul = 1; *shift = 0;
    if (cshift) *cshift = 0;
    if (op.out.buf < 0) return;
    const BufSpec& b = m->bufs[op.out.buf];
    if (b.vec) return;
    *mul = b.mul; *shift = b.shift;
    if (cshift) *cshift = b.cshift;
    if (op.kind == OP_CONV) { const ConvSpec& cs = m->convs[op.conv]; *C = cs.Cout / (op.d2s * op.d2s); }
    else if (op.kind == OP_ATTN) *C = 32;
    else if (op.kind == OP_TOBLK) *C = 64;
    else *C = b.C;
}

int sr_model_num_ops(sr_model* m) { return m ? (int)m->ops.size() : SR_ERR_INVALID; }

int sr_model_op_info(sr_model* m, int index, const char** name, int* channels, int* mul, int* shift, int* ceil_halvings) {
    if (!m || index < 0 || index >= (int)m->ops.size()) return SR_ERR_INVALID;
    if (m->op_names.size() != m->ops.size()) {
        m->op_names.clear();
        static const char* kinds[] = {"convert", "conv", "attention", "maxpool", "gap", "dense", "to_blocked", "stride2_pick", "vgg_preprocess"};
        for (const Op& op : m->ops)
            m->op_names.push_back(op.kind == OP_CONV ? m->convs[op.conv].parts[0].name : std::string(kinds[op.kind]));
    }
    int C, mu, sh, cs;
    op_out_view(m, m->ops[index], &C, &mu, &sh, &cs);
    if (name) *name = m->op_names[index].c_str();
    if (channels) *channels = C;
    if (mul) *mul = mu;
    if (shift) *shift = sh;
    if (ceil_halvings) *ceil_halvings = cs;
    return SR_OK;
}

int sr_model_set_tap(sr_model* m, int op_index, float* device_dst, int64_t capacity) {
    if (!m || op_index < 0 || op_index >= (int)m->ops.size()) return SR_ERR_INVALID;
    if (!device_dst) { m->taps.erase(op_index); return SR_OK; }
    int C, mu, sh;
    op_out_view(m, m->ops[op_index], &C, &mu, &sh);
    if (C <= 0) return m->ctx->fail(SR_ERR_INVALID, "this op has no activation-buffer output to tap");
    m->taps[op_index] = sr_model::Tap{device_dst, capacity};
    return SR_OK;
}

int sr_model_num_params(sr_model* m) { return m ? (int)m->params.size() : SR_ERR_INVALID; }

int sr_model_param_info(sr_model* m, int index, const char** name, int* which, int64_t shape[4], int* ndim) {
    if (!m || index < 0 || index >= (int)m->params.size()) return SR_ERR_INVALID;
    const Param& p = m->params[index];
    if (name) *name = p.name.c_str();
    if (which) *which = p.which;
    if (ndim) *ndim = (int)p.shape.size();
    if (shape) for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    return SR_OK;
}

int sr_model_set_weight(sr_model* m, const char* name, int which, const float* host, const int64_t* shape, int ndim) {
    if (!m || !name || !host || !shape) return SR_ERR_INVALID;
    const int i = m->find_param(name, which);
    if (i < 0) return m->ctx->fail(SR_ERR_NAME, std::string("no parameter '") + name + "' (which=" + std::to_string(which) + ")");
    Param& p = m->params[i];
    bool ok = ndim == (int)p.shape.size();
    for (int d = 0; ok && d < ndim; ++d) ok = shape[d] == p.shape[d];
    if (!ok) return m->ctx->fail(SR_ERR_INVALID, std::string("shape mismatch for '") + name + "'");
    p.host.assign(host, host + p.count());
    p.set = true;
    m->finalized = false;
    return SR_OK;
}

int sr_model_finalize(sr_model* m) {
    DeviceGuard dg_(m ? m->ctx : nullptr);
    if (!m) return SR_ERR_INVALID;
    sr_ctx* ctx = m->ctx;
    for (auto& p : m->params)
        if (!p.set) return ctx->fail(SR_ERR_STATE, "parameter '" + p.name + (p.which ? "' bias" : "' kernel") + " was never set");
    auto gather = [&](const ConvSpec& c, std::vector<float>& k, std::vector<float>& bias) {   // HWIO kernel + bias of a (possibly multi-part) conv
        const int taps = c.KS * c.KS;
        k.assign((size_t)taps * c.Cin * c.Cout, 0.f); bias.assign(c.Cout, 0.f);
        int co0 = 0;
        for (auto& part : c.parts) {
            const Param& pk = m->params[m->find_param(part.name, SR_WEIGHT_KERNEL)];
            const Param& pb = m->params[m->find_param(part.name, SR_WEIGHT_BIAS)];
            for (int t = 0; t < taps; ++t)
                for (int ci = 0; ci < c.Cin; ++ci)
                    for (int co = 0; co < part.cout; ++co)
                        k[((size_t)t * c.Cin + ci) * c.Cout + co0 + co] = part.scale * pk.host[((size_t)t * c.Cin + ci) * part.cout + co];
            for (int co = 0; co < part.cout; ++co) bias[co0 + co] = part.scale * pb.host[co];
            co0 += part.cout;
        }
    };
    for (auto& c : m->convs) {
        conv_free_weights(ctx, &c.w);
        std::vector<float> k, bias;
        gather(c, k, bias);
        int rc = conv_pack_weights(ctx, k.data(), bias.data(), c.KS, c.Cin, c.Cout, m->T, &c.w, c.rows_head);
        if (rc) return rc;
    }
    for (auto& ch : m->chains) {
        chain_free_weights(ctx, &ch.w);
        const ConvSpec& a = m->convs[ch.conv_a], &bq = m->convs[ch.conv_b];
        std::vector<float> ka, ba, kb, bb;
        gather(a, ka, ba);
        gather(bq, kb, bb);
        int rc = chain_pack_weights(ctx, ka.data(), ba.data(), kb.data(), bb.data(), a.Cin / 32, a.Cout / 16, bq.Cout / 16, &ch.w);
        if (rc) return rc;
    }
    for (auto& rt : m->rgbtails) {
        rgbtail_free_weights(ctx, &rt.w);
        const ConvSpec& c2 = m->convs[rt.conv_b];
        std::vector<float> k, bias;
        gather(c2, k, bias);
        int rc = rgbtail_pack_weights(ctx, k.data(), bias.data(), c2.Cout, &rt.w);
        if (rc) return rc;
    }
    for (auto& pj : m->projs) {
        proj_free_weights(ctx, &pj.w);
        const ConvSpec& c1 = m->convs[pj.conv];
        std::vector<float> k, bias;
        gather(c1, k, bias);
        int rc = proj_pack_weights(ctx, k.data(), bias.data(), c1.Cout, &pj.w);
        if (rc) return rc;
    }
    for (auto& pw : m->pw2s) {
        pw2_free_weights(ctx, &pw.w);
        const ConvSpec& c1 = m->convs[pw.conv];
        std::vector<float> k, bias;
        gather(c1, k, bias);
        int rc = pw2_pack_weights(ctx, k.data(), bias.data(), c1.Cin, c1.Cout, pw.act, &pw.w);
        if (rc) return rc;
    }
    for (auto& op : m->ops) {
        if (op.kind != OP_DENSE) continue;
        for (int pi : {op.dw, op.db}) {
            if (m->dense_dev[pi]) ctx->dfree(m->dense_dev[pi]);
            const Param& p = m->params[pi];
            m->dense_dev[pi] = static_cast<float*>(ctx->dalloc(sizeof(float) * p.count()));
            if (!m->dense_dev[pi]) return SR_ERR_OOM;
            SR_HIP(ctx, hipMemcpy(m->dense_dev[pi], p.host.data(), sizeof(float) * p.count(), hipMemcpyHostToDevice));
        }
    }
    m->finalized = true;
    return SR_OK;
}

int sr_model_output_shape(sr_model* m, int B, int H, int W, int C, int64_t out_shape[4]) {
    if (!m || !out_shape) return SR_ERR_INVALID;
    (void)C;
    if (m->out_vec) { out_shape[0] = B; out_shape[1] = m->out_C; out_shape[2] = 1; out_shape[3] = 1; }
    else { out_shape[0] = B; out_shape[1] = ((int64_t)H * m->out_mul) >> m->out_shift; out_shape[2] = ((int64_t)W * m->out_mul) >> m->out_shift; out_shape[3] = m->out_C; }
    return SR_OK;
}

int sr_forward(sr_model* m, const void* x, int io_dtype, int B, int H, int W, int C, void* y, int64_t y_capacity, void* stream) {
    DeviceGuard dg_(m ? m->ctx : nullptr);
    if (!m) return SR_ERR_INVALID;
    sr_ctx* ctx = m->ctx;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!m->finalized) return ctx->fail(SR_ERR_STATE, "sr_model_finalize has not been called");
    if (!x || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (io_dtype != SR_DTYPE_F32 && io_dtype != SR_DTYPE_BF16) return ctx->fail(SR_ERR_INVALID, "io dtype must be f32 or bf16");
    if (io_dtype == SR_DTYPE_BF16 && m->T == SR_DTYPE_F32) return ctx->fail(SR_ERR_INVALID, "an fp32 model takes fp32 tensors");
    if (C != m->in_C) return ctx->fail(SR_ERR_INVALID, "channel count does not match the model");
    if (B == 0) return SR_OK;   // keras predict on an empty batch returns an empty array
    if (B < 0 || H <= 0 || W <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    int64_t os[4];
    sr_model_output_shape(m, B, H, W, C, os);
    if (y_capacity < os[0] * os[1] * os[2] * os[3]) return ctx->fail(SR_ERR_CAPACITY, "output buffer too small");
    if (m->kind == SR_MODEL_VGG16 && (H < 32 || W < 32)) return ctx->fail(SR_ERR_INVALID, "VGG16 needs H,W >= 32");
    if (m->kind == SR_MODEL_VGG19_FEATURES && (H < 16 || W < 16)) return ctx->fail(SR_ERR_INVALID, "VGG19 features need H,W >= 16");
    const bool use_cells = m->taps.empty() && (ctx->chain_mask & 16) != 0;
    // 24-pixel-wide patches (patch_size_lr = 24: the reference's own training patch, ESRGAN_model.py:858 / constants.py:8) through the fused dense-block kernels, which are
    // built for 48-pixel rows: two images side by side per row, [ceil(B / 2)][H][C / 32][48][32], from the trunk's first concat buffer to trunk_conv's input; the kernels
    // treat columns 23 | 24 as an image border (dense_fused.hip SEAM).  Needs all three fused dense-block kernels (mask bits 0, 1, 5); taps read the packed buffers through tap_copy's pair mapping.
    bool pack2 = m->kind == SR_MODEL_ESRGAN_G && m->T == SR_DTYPE_BF16 && W == 24 && (ctx->chain_mask & 35) == 35 && !m->chains.empty() && B >= 2;
    for (const auto& tp : m->taps) {                                          // a tap on a tail pair's first conv makes that pair run layer by layer (below), which the packed layout cannot
        const Op& to = m->ops[tp.first];
        if (to.kind == OP_CONV && to.chain >= 0 && m->chains[to.chain].tail && to.chain_pos == 0) pack2 = false;
    }
    // Cell packing for the dense blocks on the TILE kernels (round 4): where the fused kernels do not apply (another growth width than 32, another width than 48 / two-up 24) and the
    // images fill the 16 x 16 output tiles badly -- 24 x 24: 56 % -- the concat buffers hold the batch as one image of gx cells per row, (H + 1) x (W + 1) pixels each with a zero
    // separator row / column (CellGrid): 92 % at 24 x 24.  From the trunk's first concat buffer to trunk_conv's input, as the two-up packing; the convs run on ONE tall image and never
    // store a separator.
    CellGrid cp;
    {
        static const bool no_cellpack = getenv("SR355_NO_CELLPACK") != nullptr;      // A/B switch (diagnostic)
        bool cat_blk = false;
        for (const BufSpec& bs : m->bufs) if (bs.cat && bs.blk) cat_blk = true;
        const bool fused_here = !m->chains.empty() && W == 48 && (ctx->chain_mask & 3);
        if (m->kind == SR_MODEL_ESRGAN_G && m->T == SR_DTYPE_BF16 && cat_blk && !pack2 && !fused_here && m->taps.empty() && !no_cellpack && B >= 4 && H >= 2 && W >= 2) {
            const double plain = (double)H * W / ((double)((H + 15) / 16 * 16) * ((W + 15) / 16 * 16));
            const int gx = B < 16 ? B : 16;
            const int64_t Hv = (int64_t)((B + gx - 1) / gx) * (H + 1), Wv = (int64_t)gx * (W + 1);
            const double packed = (double)B * H * W / ((double)((Hv + 15) / 16 * 16) * ((Wv + 15) / 16 * 16));
            int Cmax = 0;
            for (const BufSpec& bs : m->bufs) if (bs.cat) Cmax = std::max(Cmax, bs.Cbuf);
            if (packed >= 1.15 * plain && Hv * Wv * Cmax < ((int64_t)1 << 31)) { cp.gx = gx; cp.ch = H + 1; cp.cw = W + 1; cp.Hv = (int)Hv; cp.Wv = (int)Wv; }
        }
    }
    const int B_ws = pack2 ? (B + 1) & ~1 : cp.gx ? (int)std::max<int64_t>(B, ((int64_t)cp.Hv * cp.Wv + (int64_t)H * W - 1) / ((int64_t)H * W)) : B;
    int rc = ensure_workspace(m, B_ws, H, W, st, use_cells);
    if (rc) return rc;
    if (cp.gx) {
        // separators (and cells no image has been written to) must be zeros: adopting the grid over other contents clears the buffers; with the grid already in place
        // only the buffer the last forward unpacked trunk_conv's input into
        const bool all = !(m->cat_grid == cp);
        for (size_t i = 0; i < m->bufs.size(); ++i)
            if (m->bufs[i].cat && (all || (int)i == m->cat_dirty)) SR_HIP(ctx, hipMemsetAsync(m->bufp[i], 0, m->bufcap[i], st));
        m->cat_grid = cp;
        m->cat_dirty = -1;
    } else {
        m->cat_grid = CellGrid{};                                             // a plain or two-up forward writes where the separators were
        m->cat_dirty = -1;
    }
    SR_HIP(ctx, hipEventRecord(ctx->ev0, st));
    const int T = m->T;
    int proj_done = -1;                                                       // index of a 1x1 projection op the previous conv's epilogue has already computed
    int pool_done = -1;                                                       // index of a max-pool op the previous conv's epilogue has already computed
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        const Op& op = m->ops[oi];
        int h = H, w = W;
        if (op.in.buf >= 0) buf_hw(m->bufs[op.in.buf], H, W, &h, &w);
        switch (op.kind) {
            case OP_CONVERT:
                rc = convert_pad_launch(ctx, x, io_dtype, (int64_t)B * H * W, C, m->bufp[op.out.buf], T, m->bufs[op.out.buf].Cbuf, 1.f, 0.f, st);
                break;
            case OP_CONV: {
                const ConvSpec& cs = m->convs[op.conv];
                if ((int)oi == proj_done) break;                              // computed by the producing conv (its output buffer holds the result; a tap reads it below)
                TensorView xin{m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, m->bufs[op.in.buf].blk};
                if (op.chain >= 0) {
                    const ChainSpec& ch = m->chains[op.chain];
                    // a tap on the tail pair's first conv (conv4) needs that conv's output in memory: the fused kernel keeps it in its LDS
                    // ring, so such a pair runs layer by layer (the tap used to return stale workspace bytes without an error)
                    const size_t first = op.chain_pos == 0 ? oi : oi - 1;
                    const bool tapped_inner = ch.tail && !m->taps.empty() && m->taps.count((int)first) != 0;
                    if (!tapped_inner && (ctx->chain_mask & (ch.tail ? 1 : 2)) && chain_supported(ch.w, xin, pack2 ? 48 : w)) {
                        if (op.chain_pos == 0) {                  // the pair runs as one kernel, launched at its first op
                            const Op& ob = m->ops[oi + 1];
                            auto view = [&](const Ref& r) { return r.buf >= 0 ? TensorView{m->bufp[r.buf], m->bufs[r.buf].Cbuf, r.coff, m->bufs[r.buf].blk} : TensorView{}; };
                            TensorView outv{}, so{};
                            float bx = 0.f, bo = 0.f;
                            if (ch.tail) {
                                outv = view(ob.out);
                                for (int k2 = 0; k2 < 2; ++k2) {   // which skip is the block's own input x, which the other tensor
                                    const Ref& r = k2 ? ob.skip2 : ob.skip1;
                                    const float be = k2 ? ob.beta2 : ob.beta1;
                                    if (r.buf < 0) continue;
                                    if (r.buf == op.in.buf && r.coff == 0) bx = be; else { so = view(r); bo = be; }
                                }
                            }
                            rc = pack2 ? chain_launch(ctx, ch.w, xin, (B + 1) / 2, h, 48, outv, so, ob.alpha, bx, bo, st, true)
                                       : chain_launch(ctx, ch.w, xin, B, h, w, outv, so, ob.alpha, bx, bo, st);
                        }
                        break;
                    }
                }
                if ((ctx->chain_mask & 32) && op.chain < 0 && op.act == SR_ACT_RELU && op.alpha == 1.f && op.skip1.buf < 0 && op.skip2.buf < 0 && op.d2s == 1 && !op.clip &&
                    op.out.buf == op.in.buf && op.in.coff == 0 && op.out.coff == 64 && conv1_stream_supported(cs.w, xin, pack2 ? 48 : w)) {
                    rc = pack2 ? conv1_stream_launch(ctx, cs.w, xin, (B + 1) / 2, h, 48, st, true)
                               : conv1_stream_launch(ctx, cs.w, xin, B, h, w, st);   // conv1 of a dense block: the streaming kernel
                    break;
                }
                if (cp.gx && op.pack_alt >= 0) {
                    // trunk_conv: its input is cell-packed -- unpacked into a free concat buffer first (as for the two-up packing below)
                    rc = cell_unpack_launch(ctx, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, B, h, w, 64, m->bufp[op.pack_alt], m->bufs[op.pack_alt].Cbuf, 0, cp, st);
                    if (rc) return rc;
                    m->cat_dirty = op.pack_alt;                               // that buffer holds a plain image now: the next forward clears it before it packs
                    xin = TensorView{m->bufp[op.pack_alt], m->bufs[op.pack_alt].Cbuf, 0, m->bufs[op.pack_alt].blk};
                } else if (cp.gx && m->bufs[op.in.buf].cat && !(op.out.buf >= 0 && m->bufs[op.out.buf].cat)) {
                    return ctx->fail(SR_ERR_STATE, "cell-packed dense blocks: a conv outside them would read a packed buffer");
                }
                if (pack2 && op.pack_alt >= 0) {
                    // trunk_conv: its input, channels [0, 64) of the last dense block's buffer, is two-up packed -- unpacked into a free concat buffer first
                    rc = unpack_pairs_launch(ctx, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, B, h, w, 64, m->bufp[op.pack_alt], m->bufs[op.pack_alt].Cbuf, 0, st);
                    if (rc) return rc;
                    xin = TensorView{m->bufp[op.pack_alt], m->bufs[op.pack_alt].Cbuf, 0, m->bufs[op.pack_alt].blk};
                } else if (pack2 && m->bufs[op.in.buf].blk) {
                    return ctx->fail(SR_ERR_STATE, "two-up packed dense blocks: a conv outside the fused kernels would read a packed buffer");
                }
                if (op.rgbtail >= 0 && (ctx->chain_mask & 4) && m->taps.count((int)oi) == 0 && cs.w.rows && cs.w.NT == 4 && cs.w.Cout == 64 &&
                    op.skip1.buf < 0 && op.skip2.buf < 0 && op.d2s == 1 && !op.clip && op.act != SR_ACT_TANH && op.out.buf >= 0 &&
                    (int64_t)m->bufcap[op.out.buf] >= rgbtail_partial_bytes(B, h, w)) {
                    // final_conv1 + final_conv2 as one kernel and a finishing pass; the partial sums live in the buffer final_conv1's
                    // output would have taken (a tile's 3 KiB against the 24 KiB of its 64-channel pixels)
                    const Op& ob = m->ops[oi + 1];
                    const RgbTailSpec& rt = m->rgbtails[op.rgbtail];
                    ConvEpilogue ep1;
                    ep1.act = op.act; ep1.alpha = op.alpha;
                    ep1.f2 = &rt.w; ep1.f2_part = static_cast<float*>(m->bufp[op.out.buf]);
                    rc = conv_launch(ctx, cs.w, xin, B, h, w, TensorView{}, ep1, st);
                    if (rc) return rc;
                    rc = rgbtail_finish_launch(ctx, rt.w, ep1.f2_part, B, h, w, ob.act, ob.alpha, ob.clip, y, m->out_C, 0, io_dtype == SR_DTYPE_F32, st);
                    if (rc) return rc;
                    ++oi;                                             // the second conv has run
                    break;
                }
                if (op.pw2 >= 0 && (ctx->chain_mask & 256) && m->taps.count((int)oi) == 0 && cs.w.thin && !cs.w.few && cs.w.dtype == SR_DTYPE_F32 && op.skip1.buf < 0 &&
                    op.skip2.buf < 0 && op.d2s == 1 && !op.clip) {
                    // conv2d (9x9, ReLU) + conv2d_1 (1x1, ReLU) as one kernel: only the 32-channel tensor is stored
                    const Op& ob = m->ops[oi + 1];
                    ConvEpilogue ep2;
                    ep2.act = op.act; ep2.alpha = op.alpha; ep2.pw2 = &m->pw2s[op.pw2].w;
                    rc = conv_launch(ctx, cs.w, xin, B, h, w, TensorView{m->bufp[ob.out.buf], m->bufs[ob.out.buf].Cbuf, ob.out.coff, m->bufs[ob.out.buf].blk}, ep2, st);
                    if (rc) return rc;
                    ++oi;                                             // the 1x1 has run
                    break;
                }
                ConvEpilogue ep;
                ep.act = op.act; ep.alpha = op.alpha; ep.clip01 = op.clip; ep.d2s_r = op.d2s;
                int cB = B;
                if (op.out.buf >= 0 && m->grid_now[op.in.buf].gx) {
                    // both tensors hold the batch packed in the same CellGrid: one tall image to the kernel, separators never stored
                    const CellGrid& g = m->grid_now[op.in.buf];
                    if (!(m->grid_now[op.out.buf] == g) || op.skip1.buf >= 0 || op.skip2.buf >= 0 || op.d2s != 1)
                        return ctx->fail(SR_ERR_STATE, "packed small-image layout: producer and consumer disagree");
                    cB = 1; h = g.Hv; w = g.Wv; ep.cell_h = g.ch; ep.cell_w = g.cw;
                }
                if (cp.gx && op.out.buf >= 0 && m->bufs[op.in.buf].cat && m->bufs[op.out.buf].cat) {
                    // a dense-block conv on the cell-packed concat buffers: input, output and skips share the grid
                    if ((op.skip1.buf >= 0 && !m->bufs[op.skip1.buf].cat) || (op.skip2.buf >= 0 && !m->bufs[op.skip2.buf].cat) || op.d2s != 1)
                        return ctx->fail(SR_ERR_STATE, "cell-packed dense blocks: a skip tensor outside the grid");
                    cB = 1; h = cp.Hv; w = cp.Wv; ep.cell_h = cp.ch; ep.cell_w = cp.cw;
                }
                if (op.skip1.buf >= 0) { ep.skip1 = {m->bufp[op.skip1.buf], m->bufs[op.skip1.buf].Cbuf, op.skip1.coff, m->bufs[op.skip1.buf].blk}; ep.beta1 = op.beta1; }
                if (op.skip2.buf >= 0) { ep.skip2 = {m->bufp[op.skip2.buf], m->bufs[op.skip2.buf].Cbuf, op.skip2.coff, m->bufs[op.skip2.buf].blk}; ep.beta2 = op.beta2; }
                if ((ctx->chain_mask & 64) && T == SR_DTYPE_BF16 && oi + 1 < m->ops.size() && m->ops[oi + 1].kind == OP_POOL && m->ops[oi + 1].in.buf == op.out.buf &&
                    op.out.buf >= 0 && m->ops[oi + 1].out.buf >= 0 && cs.w.rows && cs.w.NT == 4 && cs.w.CoutP == cs.w.Cout && op.skip1.buf < 0 && op.skip2.buf < 0 &&
                    op.d2s == 1 && !op.clip && op.act != SR_ACT_TANH && m->taps.count((int)oi) == 0 && cB == B && !m->grid_now[op.out.buf].gx && h >= 2 && w >= 2 &&
                    m->bufs[m->ops[oi + 1].out.buf].Cbuf % 8 == 0 && m->bufs[m->ops[oi + 1].out.buf].Cbuf >= cs.w.Cout) {
                    // conv -> MaxPooling2D (every VGG16 block ends so): the pool rides in the conv's epilogue, the full-resolution tensor is never stored
                    const Op& opl = m->ops[oi + 1];
                    ep.pool_out = TensorView{m->bufp[opl.out.buf], m->bufs[opl.out.buf].Cbuf, opl.out.coff, m->bufs[opl.out.buf].blk};
                    ep.pool_grid = m->grid_now[opl.out.buf];
                    pool_done = (int)oi + 1;
                }
                if (op.proj >= 0 && (ctx->chain_mask & 8) && cs.w.rows && cs.w.NT == 4 && cs.w.Cout / (op.d2s * op.d2s) == 64 && cs.w.CoutP == cs.w.Cout &&
                    op.out.buf >= 0 && !m->bufs[op.out.buf].blk && op.skip2.buf < 0 && !op.clip && op.act != SR_ACT_TANH && m->bufs[op.out.buf].Cbuf % 4 == 0) {
                    // the SelfAttention layer that follows opens with three 1x1 convs of this conv's output: computed here, from registers
                    const Op& oq = m->ops[oi + 1];
                    ep.pj = &m->projs[op.proj].w;
                    ep.pj_out = TensorView{m->bufp[oq.out.buf], m->bufs[oq.out.buf].Cbuf, oq.out.coff, m->bufs[oq.out.buf].blk};
                    proj_done = (int)oi + 1;
                }
                if (op.out.buf == -2) {
                    ep.out_f32 = io_dtype == SR_DTYPE_F32;
                    rc = conv_launch(ctx, cs.w, xin, B, h, w, y, m->out_C, 0, ep, st);
                } else {
                    rc = conv_launch(ctx, cs.w, xin, cB, h, w, TensorView{m->bufp[op.out.buf], m->bufs[op.out.buf].Cbuf, op.out.coff, m->bufs[op.out.buf].blk}, ep, st);
                }
                break;
            }
            case OP_ATTN:
                rc = attention_launch(ctx, T, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, /*q=g*/ 8, /*k=f*/ 0, /*v=h*/ 16, B, h * w,
                                      m->bufp[op.out.buf], m->bufs[op.out.buf].Cbuf, 0, st);
                break;
            case OP_SUBSAMPLE:
                rc = subsample2_launch(ctx, T, m->bufp[op.in.buf], B, h, w, m->bufs[op.in.buf].C, m->bufs[op.in.buf].Cbuf, m->bufp[op.out.buf],
                                       m->bufs[op.out.buf].Cbuf, st);
                break;
            case OP_PREPROC:
                rc = vgg_preproc_launch(ctx, x, io_dtype, (int64_t)B * H * W, m->bufp[op.out.buf], T, m->bufs[op.out.buf].Cbuf, st);
                break;
            case OP_TOBLK:
                if (cp.gx) {
                    rc = cell_pack_launch(ctx, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, B, h, w, 64, m->bufp[op.out.buf], m->bufs[op.out.buf].Cbuf, op.out.coff, cp, st);
                    break;
                }
                if (pack2) {
                    rc = pack_pairs_launch(ctx, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, B, h, w, 64, m->bufp[op.out.buf], m->bufs[op.out.buf].Cbuf, op.out.coff, st);
                    break;
                }
                rc = nhwc_to_blocked_launch(ctx, m->bufp[op.in.buf], m->bufs[op.in.buf].Cbuf, op.in.coff, B, h, w, 64, m->bufp[op.out.buf],
                                            m->bufs[op.out.buf].Cbuf, op.out.coff, st);
                break;
            case OP_POOL:
                if ((int)oi == pool_done) break;                              // computed by the conv in front of it
                rc = maxpool2_launch(ctx, T, m->bufp[op.in.buf], B, h, w, m->bufs[op.in.buf].C, m->bufs[op.in.buf].Cbuf, m->bufp[op.out.buf],
                                     m->bufs[op.out.buf].Cbuf, st, m->grid_now[op.in.buf], m->grid_now[op.out.buf]);
                break;
            case OP_GAP:
                rc = gap_launch(ctx, T, m->bufp[op.in.buf], B, h * w, m->bufs[op.in.buf].C, m->bufs[op.in.buf].Cbuf,
                                static_cast<float*>(m->bufp[op.out.buf]), st);
                break;
            case OP_DENSE: {
                const float* xin = static_cast<const float*>(m->bufp[op.in.buf]);
                if (op.out.buf == -2) {
                    if (io_dtype == SR_DTYPE_F32) rc = dense_launch(ctx, xin, m->dense_dev[op.dw], m->dense_dev[op.db], B, op.In, op.Out, op.act, static_cast<float*>(y), 0, nullptr, st);
                    else rc = dense_launch(ctx, xin, m->dense_dev[op.dw], m->dense_dev[op.db], B, op.In, op.Out, op.act, nullptr, io_dtype, y, st);
                } else {
                    rc = dense_launch(ctx, xin, m->dense_dev[op.dw], m->dense_dev[op.db], B, op.In, op.Out, op.act, static_cast<float*>(m->bufp[op.out.buf]), 0, nullptr, st);
                }
                break;
            }
        }
        if (rc) return rc;
        if (!m->taps.empty()) {
            auto it = m->taps.find((int)oi);
            if (it != m->taps.end()) {
                const Op& top = m->ops[oi];                                  // oi may have moved on to a second conv computed in this one's epilogue (the fused 1x1): the tap is on THAT op
                int C_, mu, sh;
                op_out_view(m, top, &C_, &mu, &sh);
                int th, tw;
                buf_hw(m->bufs[top.out.buf], H, W, &th, &tw);
                if (it->second.cap < (int64_t)B * th * tw * C_) return ctx->fail(SR_ERR_CAPACITY, "tap buffer too small");
                const BufSpec& ob = m->bufs[top.out.buf];
                rc = tap_copy_launch(ctx, m->bufp[top.out.buf], T, ob.blk, ob.Cbuf, top.out.coff, B, th, tw, C_, it->second.dst, st, pack2 && ob.blk);
                if (rc) return rc;
            }
        }
    }
    SR_HIP(ctx, hipEventRecord(ctx->ev1, st));
    ctx->timed = true;
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------- single ops
int sr_conv2d(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int Cin, const float* w_hwio, const float* bias, int KH, int KW,
              int Cout, int act, float alpha, const void* skip1, float beta1, const void* skip2, float beta2, int clip01, int d2s_r,
              void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!x || !w_hwio || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (KH != KW) return ctx->fail(SR_ERR_INVALID, "square kernels only");
    if (dtype != SR_DTYPE_F32 && dtype != SR_DTYPE_BF16) return ctx->fail(SR_ERR_INVALID, "dtype must be f32 or bf16");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    ConvWeights cw;
    int rc = conv_pack_weights(ctx, w_hwio, bias, KH, Cin, Cout, dtype, &cw);
    if (rc) return rc;
    const int esz = dtype_size(dtype);
    const int Cp = cw.thin ? cw.CinP : round_up(cw.CinP, 32);
    void* xp = ctx->dalloc((size_t)B * H * W * Cp * esz + 4096);
    if (!xp) { conv_free_weights(ctx, &cw); return SR_ERR_OOM; }
    rc = convert_pad_launch(ctx, x, dtype, (int64_t)B * H * W, Cin, xp, dtype, Cp, 1.f, 0.f, st);
    if (!rc) {
        ConvEpilogue ep;
        ep.act = act; ep.alpha = alpha; ep.clip01 = clip01; ep.d2s_r = d2s_r < 1 ? 1 : d2s_r;
        ep.allow_splitk = 1;                                      // a single op, like sr_conv2d_dev (same kernel choice: the two agree bit for bit); model forwards do not set it
        if (skip1) { ep.skip1 = {skip1, Cout, 0}; ep.beta1 = beta1; }
        if (skip2) { ep.skip2 = {skip2, Cout, 0}; ep.beta2 = beta2; }
        const int r = ep.d2s_r;
        rc = conv_launch(ctx, cw, TensorView{xp, Cp, 0}, B, H, W, y, Cout / (r * r), 0, ep, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    ctx->dfree(xp);
    conv_free_weights(ctx, &cw);
    if (!rc && e != hipSuccess) return ctx->fail(SR_ERR_HIP, std::string("conv2d: ") + hipGetErrorString(e));
    return rc;
}

int sr_conv2d_dev(sr_ctx* ctx, const void* x, int B, int H, int W, int Cin, const float* d_w, const float* d_bias, int K, int Cout, int rot,
                  int act, float alpha, const void* skip1, float beta1, const void* skip2, float beta2, int clip01, int d2s_r, void* y,
                  void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!x || !d_w || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    ConvWeights cw;
    int rc = conv_pack_weights_dev(ctx, d_w, d_bias, K, Cin, Cout, rot, &cw, st);
    if (rc) return rc;
    const int Cp = cw.thin ? cw.CinP : round_up(cw.CinP, 32);
    const void* xp = x;
    static const bool force_copy = getenv("SR355_CONV_DEV_COPY") != nullptr;   // A/B switch (diagnostic)
    if (force_copy || Cp != Cin || ((uintptr_t)x % 16) != 0) {          // channel padding needed: a padded copy in the arena (else the conv reads x in place)
        void* xa = ctx->arena(ctx->dev_x, (size_t)B * H * W * Cp * 4 + 4096, st);
        if (!xa) return SR_ERR_OOM;
        rc = convert_pad_launch(ctx, x, SR_DTYPE_F32, (int64_t)B * H * W, Cin, xa, SR_DTYPE_F32, Cp, 1.f, 0.f, st);
        if (rc) return rc;
        xp = xa;
    }
    ConvEpilogue ep;
    ep.act = act; ep.alpha = alpha; ep.clip01 = clip01; ep.d2s_r = d2s_r < 1 ? 1 : d2s_r;
    ep.allow_splitk = 1;
    if (skip1) { ep.skip1 = {skip1, Cout, 0}; ep.beta1 = beta1; }
    if (skip2) { ep.skip2 = {skip2, Cout, 0}; ep.beta2 = beta2; }
    const int r = ep.d2s_r;
    return conv_launch(ctx, cw, TensorView{xp, Cp, 0}, B, H, W, y, Cout / (r * r), 0, ep, st);
}

// The training tape's dense blocks keep their concat tensor [B, H, W, 64 + 4 G] in ONE buffer: every conv reads a channel prefix of it and writes its
// own slice (forward), every input gradient accumulates into a prefix of the gradient buffer in place (backward) -- no concat copies, no slice copies, no
// separate accumulate launches (round 3: ~30 of them per dense block).  A view = (pointer, channels per pixel of the buffer, first channel).
int sr_conv2d_dev_views(sr_ctx* ctx, const sr_view* x, int B, int H, int W, int Cin, const float* d_w, const float* d_bias, int K, int Cout, int rot, int act, float alpha,
                        const sr_view* skip1, float beta1, const sr_view* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!x || !x->p || !d_w || !y || !y->p) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    for (const sr_view* v : {x, y, skip1})
        if (v && v->p && (v->coff < 0 || v->cs <= 0 || v->coff >= v->cs)) return ctx->fail(SR_ERR_INVALID, "bad view");
    if (x->cs - x->coff < Cin || y->cs - y->coff < Cout || (skip1 && skip1->p && skip1->cs - skip1->coff < Cout)) return ctx->fail(SR_ERR_INVALID, "a view is narrower than its channel count");
    ConvWeights cw;
    int rc = conv_pack_weights_dev(ctx, d_w, d_bias, K, Cin, Cout, rot, &cw, st);
    if (rc) return rc;
    if (cw.thin || cw.CinP != Cin) return ctx->fail(SR_ERR_INVALID, "conv views: the input channel count must be a whole number of the kernel's channel chunks (16 fp32 channels)");
    ConvEpilogue ep;
    ep.act = act; ep.alpha = alpha;
    ep.allow_splitk = 1;
    if (skip1 && skip1->p) { ep.skip1 = {skip1->p, skip1->cs, skip1->coff}; ep.beta1 = beta1; }
    return conv_launch(ctx, cw, TensorView{x->p, x->cs, x->coff}, B, H, W, TensorView{y->p, y->cs, y->coff}, ep, st);
}

int sr_conv_prepack(sr_ctx* ctx, const sr_pack_desc* uses, int n, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (n < 0 || (n > 0 && !uses)) { ctx->pack_cache.clear(); return ctx->fail(SR_ERR_INVALID, "conv prepack: bad list"); }
    return conv_prepack_dev(ctx, uses, n, static_cast<hipStream_t>(stream));
}

int sr_conv2d_wgrad_views(sr_ctx* ctx, const sr_view* x, const sr_view* dy, int B, int H, int W, int Cin, int Cout, int K, float* dw_hwio, float* db, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !x->p || !dy || !dy->p || !dw_hwio) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (x->coff < 0 || dy->coff < 0 || x->cs - x->coff < Cin || dy->cs - dy->coff < Cout) return ctx->fail(SR_ERR_INVALID, "a view is narrower than its channel count");
    return wgrad_launch_views(ctx, static_cast<const float*>(x->p) + x->coff, x->cs, static_cast<const float*>(dy->p) + dy->coff, dy->cs, B, H, W, Cin, Cout, K, dw_hwio, db,
                              static_cast<hipStream_t>(stream));
}

int sr_eltwise_views(sr_ctx* ctx, int op, const sr_view* a, const sr_view* b, float alpha, float beta, const sr_view* out, int64_t npix, int C, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !a->p || !out || !out->p) return ctx->fail(SR_ERR_INVALID, "null tensor");
    for (const sr_view* v : {a, b, out})
        if (v && v->p && (v->coff < 0 || v->cs - v->coff < C)) return ctx->fail(SR_ERR_INVALID, "a view is narrower than its channel count");
    const float* bp = (b && b->p) ? static_cast<const float*>(b->p) + b->coff : nullptr;
    return eltwise_views_launch(ctx, op, static_cast<const float*>(a->p) + a->coff, a->cs, bp, bp ? b->cs : 0, alpha, beta,
                                static_cast<float*>(const_cast<void*>(out->p)) + out->coff, out->cs, npix, C, static_cast<hipStream_t>(stream));
}

int sr_self_attention(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, const float* wf, const float* bf, const float* wg,
                      const float* bg, const float* wh, const float* bh, const float* wv, const float* bv, void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C != 64) return ctx->fail(SR_ERR_INVALID, "SelfAttention kernel is built for channels=64 (d_qk=8, d_v=32)");
    if (dtype != SR_DTYPE_F32 && dtype != SR_DTYPE_BF16) return ctx->fail(SR_ERR_INVALID, "dtype must be f32 or bf16");
    if (B <= 0 || H <= 0 || W <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    // f | g | h side by side: one 1x1 conv 64 -> 48
    const float kscale = dtype == SR_DTYPE_BF16 ? 1.4426950408889634f : 1.f;   // see Builder::self_attention
    std::vector<float> wp((size_t)64 * 48), bp(48);
    for (int ci = 0; ci < 64; ++ci) {
        for (int co = 0; co < 8; ++co) { wp[(size_t)ci * 48 + co] = kscale * wf[ci * 8 + co]; wp[(size_t)ci * 48 + 8 + co] = wg[ci * 8 + co]; }
        for (int co = 0; co < 32; ++co) wp[(size_t)ci * 48 + 16 + co] = wh[ci * 32 + co];
    }
    for (int co = 0; co < 8; ++co) { bp[co] = bf ? kscale * bf[co] : 0.f; bp[8 + co] = bg ? bg[co] : 0.f; }
    for (int co = 0; co < 32; ++co) bp[16 + co] = bh ? bh[co] : 0.f;
    ConvWeights cp, cv;
    int rc = conv_pack_weights(ctx, wp.data(), bp.data(), 1, 64, 48, dtype, &cp);
    if (rc) return rc;
    rc = conv_pack_weights(ctx, wv, bv, 1, 32, 64, dtype, &cv);
    if (rc) { conv_free_weights(ctx, &cp); return rc; }
    const int esz = dtype_size(dtype);
    const int64_t npix = (int64_t)B * H * W;
    void* qkv = ctx->dalloc((size_t)npix * 64 * esz + 4096);
    void* ao = ctx->dalloc((size_t)npix * 32 * esz + 4096);
    if (!qkv || !ao) rc = SR_ERR_OOM;
    if (!rc) rc = (hipMemsetAsync(qkv, 0, (size_t)npix * 64 * esz, st) == hipSuccess) ? SR_OK : ctx->fail(SR_ERR_HIP, "memset");
    ConvEpilogue e0;
    if (!rc) rc = conv_launch(ctx, cp, TensorView{x, 64, 0}, B, H, W, qkv, 64, 0, e0, st);
    if (!rc) rc = attention_launch(ctx, dtype, qkv, 64, 8, 0, 16, B, H * W, ao, 32, 0, st);
    ConvEpilogue e1;
    e1.skip1 = {x, 64, 0}; e1.beta1 = 1.f;
    if (!rc) rc = conv_launch(ctx, cv, TensorView{ao, 32, 0}, B, H, W, y, 64, 0, e1, st);
    hipError_t e = hipStreamSynchronize(st);
    ctx->dfree(qkv); ctx->dfree(ao);
    conv_free_weights(ctx, &cp); conv_free_weights(ctx, &cv);
    if (!rc && e != hipSuccess) return ctx->fail(SR_ERR_HIP, std::string("self_attention: ") + hipGetErrorString(e));
    return rc;
}

int sr_bicubic(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW, void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return bicubic_launch(ctx, x, dtype, B, H, W, C, outH, outW, y, dtype, C, static_cast<hipStream_t>(stream));
}

int sr_resize(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C, int outH, int outW, int interpolation, void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return resize_launch(ctx, x, dtype, B, H, W, C, outH, outW, interpolation, y, static_cast<hipStream_t>(stream));
}

int sr_psnr(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float max_val, float* out_B, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !out_B) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return psnr_launch(ctx, static_cast<const float*>(a), static_cast<const float*>(b), B, (int64_t)H * W * C, max_val, out_B,
                       static_cast<hipStream_t>(stream));
}

int sr_ssim(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float max_val, float* out_B, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !out_B) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return ssim_launch(ctx, static_cast<const float*>(a), static_cast<const float*>(b), B, H, W, C, max_val, out_B,
                       static_cast<hipStream_t>(stream));
}

int sr_mse(sr_ctx* ctx, const void* a, const void* b, int64_t n, float* out1, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !out1) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return mse_launch(ctx, static_cast<const float*>(a), static_cast<const float*>(b), n, out1, static_cast<hipStream_t>(stream));
}

int sr_conv2d_wgrad(sr_ctx* ctx, const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, int K, float* dw_hwio, float* db, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !dy || !dw_hwio) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return wgrad_launch(ctx, static_cast<const float*>(x), static_cast<const float*>(dy), B, H, W, Cin, Cout, K, dw_hwio, db, static_cast<hipStream_t>(stream));
}

int sr_eltwise(sr_ctx* ctx, int op, const void* a, const void* b, float alpha, float beta, void* out, int64_t n, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !out) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return eltwise_launch(ctx, op, static_cast<const float*>(a), static_cast<const float*>(b), alpha, beta, static_cast<float*>(out), n, static_cast<hipStream_t>(stream));
}

int sr_adam(sr_ctx* ctx, void* w, const void* g, void* m, void* v, int64_t n, float lr_t, float beta1, float one_minus_beta1, float beta2,
            float one_minus_beta2, float epsilon, float grad_scale, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!w || !g || !m || !v) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return adam_launch(ctx, static_cast<float*>(w), static_cast<const float*>(g), static_cast<float*>(m), static_cast<float*>(v), n, lr_t, beta1,
                       one_minus_beta1, beta2, one_minus_beta2, epsilon, grad_scale, static_cast<hipStream_t>(stream));
}

int sr_space_to_depth(sr_ctx* ctx, const void* x, int B, int H, int W, int C, int r, void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return space_to_depth_launch(ctx, static_cast<const float*>(x), B, H, W, C, r, static_cast<float*>(y), static_cast<hipStream_t>(stream));
}

int sr_spatial_op(sr_ctx* ctx, int op, const void* x, int B, int H, int W, int C, void* y, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !y) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return ctx->fail(SR_ERR_INVALID, "bad tensor shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (op) {
        case SR_SP_MAXPOOL2: return maxpool2_launch(ctx, SR_DTYPE_F32, x, B, H, W, C, C, y, C, st);
        case SR_SP_GAP: return gap_launch(ctx, SR_DTYPE_F32, x, B, H * W, C, C, static_cast<float*>(y), st);
        case SR_SP_PICK2: return subsample2_launch(ctx, SR_DTYPE_F32, x, B, H, W, C, C, y, C, st);
        case SR_SP_VGG_PREPROCESS:
            if (C != 3) return ctx->fail(SR_ERR_INVALID, "VGG preprocessing takes RGB");
            return vgg_preproc_launch(ctx, x, SR_DTYPE_F32, (int64_t)B * H * W, y, SR_DTYPE_F32, 3, st);
    }
    return ctx->fail(SR_ERR_INVALID, "unknown spatial op");
}

#define SR_F(p) static_cast<const float*>(p)
#define SR_FM(p) static_cast<float*>(p)
int sr_matmul(sr_ctx* ctx, const void* A, const void* B, void* C, int batch, int M, int N, int K, int transA, int transB, float alpha, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!A || !B || !C) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return matmul_launch(ctx, SR_F(A), SR_F(B), SR_FM(C), batch, M, N, K, transA, transB, alpha, static_cast<hipStream_t>(stream));
}
int sr_softmax_rows(sr_ctx* ctx, void* s, int64_t rows, int cols, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!s) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return softmax_rows_launch(ctx, SR_FM(s), rows, cols, static_cast<hipStream_t>(stream));
}
int sr_softmax_bwd(sr_ctx* ctx, const void* p, const void* dp, void* ds, int64_t rows, int cols, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!p || !dp || !ds) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return softmax_bwd_launch(ctx, SR_F(p), SR_F(dp), SR_FM(ds), rows, cols, static_cast<hipStream_t>(stream));
}
int sr_maxpool2_bwd(sr_ctx* ctx, const void* x, const void* dy, int B, int H, int W, int C, void* dx, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!x || !dy || !dx) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return maxpool2_bwd_launch(ctx, SR_F(x), SR_F(dy), B, H, W, C, SR_FM(dx), static_cast<hipStream_t>(stream));
}
int sr_zero_insert2(sr_ctx* ctx, const void* dy, int B, int H, int W, int C, void* out, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!dy || !out) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return zero_insert2_launch(ctx, SR_F(dy), B, H, W, C, SR_FM(out), static_cast<hipStream_t>(stream));
}
int sr_spectral_l1_bwd(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float scale, void* da, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !da) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return spectral_l1_bwd_launch(ctx, SR_F(a), SR_F(b), B, H, W, C, scale, SR_FM(da), static_cast<hipStream_t>(stream));
}

int sr_l1(sr_ctx* ctx, const void* a, const void* b, int64_t n, float* out1, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !out1) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return l1_launch(ctx, static_cast<const float*>(a), static_cast<const float*>(b), n, out1, static_cast<hipStream_t>(stream));
}

int sr_spectral_l1(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float* out1, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!a || !b || !out1) return ctx->fail(SR_ERR_INVALID, "null tensor");
    return spectral_l1_launch(ctx, static_cast<const float*>(a), static_cast<const float*>(b), B, H, W, C, out1, static_cast<hipStream_t>(stream));
}

int sr_extract_patches(sr_ctx* ctx, const float* img, int H, int W, int C, int patch, int stride, float mul, float add, int out_dtype,
                       void* out, int64_t out_capacity, int* n_patches, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (H <= 0 || W <= 0 || C <= 0 || patch <= 0 || stride <= 0) return ctx->fail(SR_ERR_INVALID, "bad shape/patch/stride");
    const int ph = pad_amount(H, patch, stride), pw = pad_amount(W, patch, stride);
    if (ph >= H || pw >= W) return ctx->fail(SR_ERR_INVALID, "reflect padding needs pad < image size");   // np.pad would wrap repeatedly
    const int Hp = H + ph, Wp = W + pw;
    const int ny = Hp >= patch ? (Hp - patch) / stride + 1 : 0, nx = Wp >= patch ? (Wp - patch) / stride + 1 : 0;
    if (n_patches) *n_patches = ny * nx;
    if (!out) return SR_OK;
    if (!img) return ctx->fail(SR_ERR_INVALID, "null image");
    if (out_dtype != SR_DTYPE_F32 && out_dtype != SR_DTYPE_BF16) return ctx->fail(SR_ERR_INVALID, "out dtype must be f32 or bf16");
    if (out_capacity < (int64_t)ny * nx * patch * patch * C) return ctx->fail(SR_ERR_CAPACITY, "patch buffer too small");
    return extract_patches_launch(ctx, img, H, W, C, patch, stride, mul, add, out_dtype, out, ny, nx, static_cast<hipStream_t>(stream));
}

int sr_overlap_add(sr_ctx* ctx, const void* patches, int in_dtype, int H, int W, int C, int patch, int stride, int scale, float mul,
                   float add, float* out, void* stream) {
    DeviceGuard dg_(ctx);
    if (!ctx) return SR_ERR_INVALID;
    if (!patches || !out) return ctx->fail(SR_ERR_INVALID, "null tensor");
    if (H <= 0 || W <= 0 || C <= 0 || patch <= 0 || stride <= 0 || scale <= 0) return ctx->fail(SR_ERR_INVALID, "bad shape/patch/stride");
    if (in_dtype != SR_DTYPE_F32 && in_dtype != SR_DTYPE_BF16) return ctx->fail(SR_ERR_INVALID, "patch dtype must be f32 or bf16");
    const int Hp = H + pad_amount(H, patch, stride), Wp = W + pad_amount(W, patch, stride);
    const int ny = (Hp - patch) / stride + 1, nx = (Wp - patch) / stride + 1;
    return overlap_add_launch(ctx, patches, in_dtype, H, W, C, patch, stride, scale, mul, add, ny, nx, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
