/* sr355.h -- C ABI of libsr355.so: the MI355X (gfx950) super-resolution + defect-classifier hot path.
 *
 * The reference (bgmanuel99/Super-Resolution-Images-for-3D-Printing-Defect-Detection) has no
 * FFI/plugin interface; its boundary to the device is the Keras call `model.predict(...)` /
 * `generator(...)` / `tf.image.psnr|ssim` / `cv2.resize`.  Each entry point below names the
 * reference call site (file:line under /root/reference) it stands in for.  The Python host
 * (package sr355, mirroring the reference's SRModels/ classes) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *  - return 0 (SR_OK) on success, negative sr_status on error; text via sr_last_error(ctx).
 *  - nothing throws across the ABI.
 *  - image tensors are NHWC, dense, DEVICE pointers owned by the caller (e.g. torch tensors'
 *    data_ptr()); conv kernels handed to sr_model_set_weight are HOST float32 HWIO, Dense [in,out].
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream) and
 *    is asynchronous; the caller synchronises.
 *  - one sr_ctx per GPU; a ctx (and its models) is not thread-safe; different ctxs are independent.
 */
#ifndef SR355_H
#define SR355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sr_ctx sr_ctx;
typedef struct sr_model sr_model;

typedef enum sr_status {
    SR_OK = 0,
    SR_ERR_INVALID = -1,   /* bad argument / unsupported shape                 */
    SR_ERR_HIP = -2,       /* HIP runtime error (message holds hipGetErrorString) */
    SR_ERR_OOM = -3,       /* device allocation failed                          */
    SR_ERR_STATE = -4,     /* model not finalised / weight missing              */
    SR_ERR_NAME = -5,      /* unknown layer name                                */
    SR_ERR_CAPACITY = -6   /* caller's output buffer too small                  */
} sr_status;

enum { SR_DTYPE_F32 = 0, SR_DTYPE_BF16 = 1, SR_DTYPE_U8 = 2 };
enum { SR_MODEL_SRCNN = 0, SR_MODEL_EDSR = 1, SR_MODEL_ESRGAN_G = 2, SR_MODEL_VGG16 = 3,
       SR_MODEL_ESRGAN_D = 4,          /* discriminator, inference graph (ESRGAN_model.py:347-377): [B,H,W,3] in [-1,1] -> [B,1] probabilities */
       SR_MODEL_VGG19_FEATURES = 5 };  /* perceptual-loss extractor incl. preprocessing (ESRGAN_model.py:379-408): -> [B,H/16,W/16,512]     */
enum { SR_ACT_LINEAR = 0, SR_ACT_RELU = 1, SR_ACT_LRELU = 2, SR_ACT_TANH = 3 };
enum { SR_WEIGHT_KERNEL = 0, SR_WEIGHT_BIAS = 1 };
/* sr_eltwise ops: out = alpha*a + beta*b | dy where y > 0 | dy (0.2 dy where y <= 0) | dy where 0 <= x <= 1 | alpha*a*b | dy*(1 - y^2) |
 * clip(a, 0, 1) | alpha * sign(a - b) */
enum { SR_ELT_AXPBY = 0, SR_ELT_RELU_BWD = 1, SR_ELT_LRELU_BWD = 2, SR_ELT_CLIP01_BWD = 3, SR_ELT_MUL = 4, SR_ELT_TANH_BWD = 5, SR_ELT_CLIP01 = 6,
       SR_ELT_SIGN_DIFF = 7 };

/* Architecture hyper-parameters = the keyword arguments of the reference's setup_model():
 * SRCNN_model.py:23, EDSR_model.py:29, ESRGAN_model.py:108, VGG16_model.py:21. */
typedef struct sr_model_cfg {
    int32_t compute_dtype;    /* SR_DTYPE_F32 (reference precision) or SR_DTYPE_BF16 (fp32 accumulate) */
    int32_t scale_factor;     /* EDSR: 2,3,4; ESRGAN: 2,4,8; ignored otherwise                         */
    int32_t channels;         /* image channels (3)                                                    */
    int32_t num_blocks;       /* EDSR num_res_blocks / ESRGAN num_rrdb_blocks                          */
    int32_t num_filters;      /* EDSR num_filters (ESRGAN trunk is fixed at 64)                        */
    int32_t growth_channels;  /* ESRGAN growth_channels                                                */
    float   res_scaling;      /* EDSR res_scaling                                                      */
    int32_t num_classes;      /* VGG16 classifier head width                                           */
    int32_t use_attention;    /* ESRGAN: 1 = reference graph (two SelfAttention layers); 0 = tuning-only variant */
} sr_model_cfg;

/* ---- context ------------------------------------------------------------------------------ */
int  sr_init(int device_id, sr_ctx** out);
void sr_destroy(sr_ctx* ctx);
const char* sr_last_error(sr_ctx* ctx);
/* bytes of device memory the library currently holds / has held at most (weights + workspaces).
 * Feeds the reference's inference_metrics keys gpu_mean_current_mb / gpu_peak_mb
 * (SRCNN_model.py:201-242, tf.config.experimental.get_memory_info). */
int  sr_mem_info(sr_ctx* ctx, int64_t* current_bytes, int64_t* peak_bytes);
/* wall-clock of the last sr_forward on this ctx measured with HIP events on its stream (ms);
 * blocks until that forward has finished. */
int  sr_last_forward_ms(sr_ctx* ctx, float* ms);

/* Shader clock (MHz) the GPU holds under a dense bf16 MFMA load, measured in-kernel (s_memtime / s_memrealtime around an
 * MFMA loop, ~0.3 s of load first).  bench.py prices the MFMA roof at this clock beside the nominal peak -- the reference
 * has no counterpart (it never names its hardware).  Blocks until measured. */
int  sr_measure_clock(sr_ctx* ctx, float* mhz, void* stream);

/* Per-launch timing of the hot kernels with HIP events recorded on the launching stream (the
 * reference's analogue is its time.perf_counter() bracket around predict, SRCNN_model.py:208-212).
 * begin: start collecting; end: synchronise the device and write a JSON array
 * [{"kernel","launches","total_ms","flops","bytes"}...] (algorithmic FLOP / HBM bytes per kernel
 * template instance) into `json` (capacity `cap` bytes). */
/* Diagnostic only (never set in production): a device buffer of 16 uint64 per workgroup of the next 3x3 bf16 conv
 * launches; a separately compiled stamped variant of the kernel writes s_memtime stamps there (NULL switches back).
 * `capacity_bytes` is the size of that buffer: a launch whose grid would write beyond it is refused with SR_ERR_INVALID
 * instead of running (round 2's one memory fault was a probe handing a buffer sized for another kernel).
 * sr_debug_stamp_bytes_needed(0, workgroups) is the size a launch of `workgroups` workgroups needs. */
int  sr_debug_set_stamp_buffer(sr_ctx* ctx, void* device_u64_buffer, int64_t capacity_bytes);
/* Diagnostic only: a device buffer of 64 x 4 x 64 x 4 uint64 for the stamped variant of the fused dense-block kernels (s_memtime at four
 * points of each of the first 64 granules, waves 0, 5, 8, 11 of the first 64 workgroups; tools/probe_chain.py).  NULL switches back.
 * A non-NULL buffer smaller than sr_debug_stamp_bytes_needed(1, 0) is refused with SR_ERR_INVALID. */
int  sr_debug_set_chain_stamp_buffer(sr_ctx* ctx, void* device_u64_buffer, int64_t capacity_bytes);
/* Bytes of stamp buffer the stamped diagnostic kernels write: which = 0 -> the 3x3 conv (16 uint64 per workgroup of the launch),
 * which = 1 -> the fused dense-block kernels (fixed size; `workgroups` ignored).  Pure function, no context; -1 for anything else. */
int64_t sr_debug_stamp_bytes_needed(int which, int64_t workgroups);
/* Which dense-block conv pairs of the ESRGAN trunk run as one fused line-buffer kernel when the shape allows (bf16, 32 growth
 * channels, images 48 pixels wide, or 24 wide with two images per 48-pixel row when bits 0, 1 and 5 are all set): bit 0 = conv4+conv5, bit 1 = conv2+conv3; bit 2 = the generator's last conv (64 -> image channels,
 * ESRGAN_model.py:341) computed inside final_conv1's epilogue, so that final_conv1's 64-channel output is never stored; bit 3 = the three
 * 1x1 projections that open a SelfAttention layer (ESRGAN_model.py:48-56) computed in the epilogue of the conv producing its input; bit 4 = batches of small
 * images (the VGG16 classifier's block 5: 6 x 6 pixels for 96-pixel patches) packed side by side with zero separator rows / columns into one
 * tall image for the 3x3 kernel, whose 12 x 16 output tiles a single such image would fill to 19 %; bit 5 = conv1 of a dense block (64 -> 32 channels, the block's one
 * HBM-bound conv) on a streaming line-buffer kernel with its weights resident in LDS; bit 6 = a MaxPooling2D computed in the epilogue of the conv in front of it
 * (every VGG16 block ends conv -> pool: the full-resolution tensor is never stored); bit 7 = 3x3 convs from 64 input channels (the generator's
 * up-sampling convs, EDSR's body) on a persistent kernel that keeps a 64-cout tile's weights in LDS; bit 8 = SRCNN's 1x1 conv (conv2d_1, 96 -> 32, ReLU:
 * SRCNN_model.py:51) computed from the accumulators in the epilogue of the 9x9 head, whose 96-channel fp32 output is then never stored; default 511.
 * mask 0 = layer by layer (the A/B switch of the parity tests and of tools/ benchmarks).  max_workgroups > 0 caps the persistent grid (tests: several images per workgroup
 * at small batches); 0 = one workgroup per CU. */
int  sr_debug_set_fused(sr_ctx* ctx, int mask, int max_workgroups);
/* Test hook: device allocations through this ctx fail (SR_ERR_OOM) once the bytes it holds would exceed `bytes`
 * (0 = no cap).  Lets the tests walk the out-of-memory path of sr_forward without filling a 288 GB card. */
int  sr_debug_set_alloc_cap(sr_ctx* ctx, int64_t bytes);
int  sr_profile_begin(sr_ctx* ctx);
int  sr_profile_end(sr_ctx* ctx, char* json, int64_t cap);

/* ---- models: replaces Keras model build + predict ------------------------------------------ */
/* Keras graph construction: SRCNN_model.py:45-53, EDSR_model.py:96-125, ESRGAN_model.py:303-345,
 * VGG16_model.py:57-97. */
int  sr_model_create(sr_ctx* ctx, int kind, const sr_model_cfg* cfg, sr_model** out);
void sr_model_destroy(sr_model* m);
/* number of parameter tensors the graph expects, and the i-th one's Keras layer name, which
 * (kernel/bias) and shape (ndim <= 4).  Lets the host enumerate what load_model() would restore. */
int  sr_model_num_params(sr_model* m);
int  sr_model_param_info(sr_model* m, int index, const char** name, int* which, int64_t shape[4], int* ndim);
/* copy one parameter (host fp32; conv HWIO, dense [in,out], bias [out]) into the model.
 * Stands in for keras load_model / set_weights (SRCNN_model.py:35, ESRGAN_model.py:143-149). */
int  sr_model_set_weight(sr_model* m, const char* keras_layer_name, int which,
                         const float* host, const int64_t* shape, int ndim);
/* pack all weights into the device MFMA layouts; must precede sr_forward. */
int  sr_model_finalize(sr_model* m);
/* Free the model's activation workspaces (they are grow-only otherwise; weights stay).  sr_forward releases them by
 * itself when growing them fails with SR_ERR_OOM, so a retry with a smaller batch starts clean -- the analogue of
 * the reference's tf.keras.backend.clear_session() between runs (defect_detection_pipeline / notebooks). */
int  sr_model_release_workspace(sr_model* m);
/* Diagnostics for stage-by-stage parity traces (tests/): the graph as a flat op list -- name = Keras layer name of a
 * conv (ESRGAN_model.py:230-341), else the op kind; the op's output is [B, h, w, channels] with h = (H*mul)>>shift then
 * ceil-halved ceil_halvings times (stride-2 SAME convs), w likewise (channels 0: no tensor output) -- and a tap that copies op `op_index`'s output, as dense fp32 NHWC, into
 * `device_dst` during every later sr_forward (NULL removes the tap).  Never set in production runs. */
int  sr_model_num_ops(sr_model* m);
int  sr_model_op_info(sr_model* m, int index, const char** name, int* channels, int* mul, int* shift, int* ceil_halvings);
int  sr_model_set_tap(sr_model* m, int op_index, float* device_dst, int64_t capacity);
/* output shape for an input of [B,H,W,C]. */
int  sr_model_output_shape(sr_model* m, int B, int H, int W, int C, int64_t out_shape[4]);
/* model.predict / generator(x): SRCNN_model.py:210, EDSR_model.py:274, ESRGAN_model.py:941,
 * VGG16_model.py:244.  x,y DEVICE NHWC of `io_dtype` (f32 or bf16; VGG16 output is [B,num_classes]).
 * y_capacity in elements. */
int  sr_forward(sr_model* m, const void* x, int io_dtype, int B, int H, int W, int C,
                void* y, int64_t y_capacity, void* stream);

/* ---- single ops (also used by the kernel-level parity tests) -------------------------------- */
/* Keras Conv2D(padding="same", strides=1) + bias + activation, then
 * y = alpha*act(conv+b) + beta1*skip1 + beta2*skip2, optional clip[0,1], optional depth_to_space(r)
 * in TF "DCR" order (EDSR_model.py:61-90, ESRGAN_model.py:230-299).  x [B,H,W,Cin], w HOST HWIO,
 * skips [B,H,W,Cout] (same dtype as x), y [B,H*r,W*r,Cout/r^2].  dtype applies to x, skips and y. */
int  sr_conv2d(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int Cin,
               const float* w_hwio, const float* bias, int KH, int KW, int Cout, int act,
               float alpha, const void* skip1, float beta1, const void* skip2, float beta2,
               int clip01, int d2s_r, void* y, void* stream);
/* SelfAttention.call (ESRGAN_model.py:48-70) on x [B,H,W,C]; weights HOST HWIO 1x1. */
int  sr_self_attention(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C,
                       const float* wf, const float* bf, const float* wg, const float* bg,
                       const float* wh, const float* bh, const float* wv, const float* bv,
                       void* y, void* stream);
/* cv2.resize(..., interpolation=cv2.INTER_CUBIC) (classic_algorithms.py:11-13, SRCNN_model.py:191,
 * loading_methods.py:147).  dtype f32: float path; u8: OpenCV fixed-point path. */
int  sr_bicubic(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C,
                int outH, int outW, void* y, void* stream);
/* cv2.resize(x, (outW, outH), interpolation=code) with OpenCV's codes INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2,
 * INTER_AREA = 3, INTER_LANCZOS4 = 4: classic_algorithms.py:7-21 (interpolate_bilinear / _bicubic / _area / _lanczos) and the
 * per-file codes of interpolation_map.pkl in load_dataset_as_patches(mode="srcnn") (loading_methods.py:131-148, which hands an
 * integer entry of the map to cv2.resize as it is).  f32: float path; u8: OpenCV's 11-bit fixed-point path; uint8 INTER_AREA
 * shrinking: integer cell sums for whole-number factors ((sum + 2) >> 2 for 2 x 2, a rounded float product otherwise), the
 * float taps rounded half to even for the others.  Any other code: SR_ERR_INVALID. */
int  sr_resize(sr_ctx* ctx, const void* x, int dtype, int B, int H, int W, int C,
               int outH, int outW, int interpolation, void* y, void* stream);
/* tf.image.psnr / tf.image.ssim(max_val) (metrics.py:3-7): a,b f32 [B,H,W,C] -> out f32 [B] (device). */
int  sr_psnr(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float max_val,
             float* out_B, void* stream);
int  sr_ssim(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float max_val,
             float* out_B, void* stream);
/* mean squared error over all elements -> out f32 [1] (Keras loss="mean_squared_error",
 * SRCNN_model.py:59). */
int  sr_mse(sr_ctx* ctx, const void* a, const void* b, int64_t n, float* out1, void* stream);
/* Pieces of the ESRGAN generator loss (ESRGAN_model.py:433-473; _train_step :511-523, evaluate :812-826): a, b f32 [B,H,W,C] device.
 * sr_l1: mean |a - b| over n elements (_pixel_loss).  sr_spectral_l1: mean | |F(a)| - |F(b)| | with tf.signal.fft2d's axes, the
 * INNERMOST two of NHWC = (W, C) (_spectral_loss; C must be 3).  Both write one float.  The adversarial term is a [B,1] vector
 * (host arithmetic), the perceptual term is sr_mse on two SR_MODEL_VGG19_FEATURES outputs. */
int  sr_l1(sr_ctx* ctx, const void* a, const void* b, int64_t n, float* out1, void* stream);
int  sr_spectral_l1(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float* out1, void* stream);
/* Backward-pass pieces of Keras model.fit (SRCNN_model.py:84-90, EDSR_model.py:164-170: loss = mean_squared_error, Adam) and of
 * ESRGAN._train_step (ESRGAN_model.py:475-533).  fp32 device tensors, NHWC dense.
 * sr_conv2d_wgrad: gradient of a Keras Conv2D(K x K, SAME, stride 1) kernel and bias: dw HWIO [K,K,Cin,Cout] = sum over pixels of
 *   x[b, y+ky-p, x+kx-p, ci] * dy[b,y,x,co];  db [Cout] = sum of dy (NULL to skip).  The input gradient needs no entry point of its
 *   own: it is sr_conv2d on dy with the kernel rotated by 180 degrees and its channel axes swapped.
 * sr_eltwise: the element-wise halves of the chain rule (SR_ELT_*), a / b / out of n floats (b may be NULL for AXPBY with beta 0).
 * sr_space_to_depth: the inverse of tf.nn.depth_to_space (DCR order): x [B,H*r,W*r,C] -> y [B,H,W,r*r*C] (its gradient). */
int  sr_conv2d_wgrad(sr_ctx* ctx, const void* x, const void* dy, int B, int H, int W, int Cin, int Cout, int K,
                     float* dw_hwio, float* db, void* stream);
/* sr_conv2d with DEVICE fp32 weights (the training loop keeps its parameters on the device between optimiser steps): d_w is a device
 * HWIO tensor [K,K,Cin,Cout]; with rot = 1 it is instead the forward kernel [K,K,Cout,Cin] of the layer whose INPUT gradient is wanted,
 * and the conv runs on its 180-degree-rotated, channel-swapped form (ESRGAN_model.py:505-533 via tf.GradientTape).  The kernel is packed
 * into MFMA fragment order by a device kernel; fully asynchronous on `stream`; scratch is reused call to call in stream order, so one
 * stream per context.  fp32 tensors only. */
int  sr_conv2d_dev(sr_ctx* ctx, const void* x, int B, int H, int W, int Cin, const float* d_w, const float* d_bias, int K, int Cout,
                   int rot, int act, float alpha, const void* skip1, float beta1, const void* skip2, float beta2, int clip01,
                   int d2s_r, void* y, void* stream);
/* The same three pieces of the training step on CHANNEL RANGES of NHWC fp32 buffers (ESRGAN_model.py:212-254: a dense block's concat tensor kept in one buffer,
 * every conv reading a prefix of it and writing its own slice; in the backward pass every input gradient accumulating in place into a prefix of the gradient
 * buffer through skip1 = y).  sr_view: p = the buffer, cs = its channels per pixel, coff = the view's first channel (cs, coff multiples of 4; the conv's Cin a
 * multiple of 16).  sr_eltwise_views: op over npix pixels x C channels. */
typedef struct { const void* p; int64_t cs; int32_t coff; } sr_view;
int  sr_conv2d_dev_views(sr_ctx* ctx, const sr_view* x, int B, int H, int W, int Cin, const float* d_w, const float* d_bias, int K, int Cout, int rot,
                         int act, float alpha, const sr_view* skip1, float beta1, const sr_view* y, void* stream);
int  sr_conv2d_wgrad_views(sr_ctx* ctx, const sr_view* x, const sr_view* dy, int B, int H, int W, int Cin, int Cout, int K,
                           float* dw_hwio, float* db, void* stream);
int  sr_eltwise_views(sr_ctx* ctx, int op, const sr_view* a, const sr_view* b, float alpha, float beta, const sr_view* out, int64_t npix, int C, void* stream);
/* Pack the weights of n conv uses by ONE launch, now, on `stream`, from the current contents of w / bias (a training step otherwise packs once per sr_conv2d_dev /
 * _views call: ~800 launches of ~5 us in the step of ESRGAN_model.py:475-533).  A later sr_conv2d_dev / sr_conv2d_dev_views on this context whose (d_w, d_bias, K, Cin,
 * Cout, rot) equals a listed use takes the pack made here instead of packing again; Cin / Cout are those of the CALL (for rot = 1, the gradient's channels in, the
 * layer's input channels out).  Every call replaces the whole list; n = 0 forgets it.  The caller must call again after changing any listed weight (a trainer: at the
 * start of every step) -- the library cannot see a write to w. */
typedef struct { const float* w; const float* bias; int32_t K, Cin, Cout, rot; } sr_pack_desc;
int  sr_conv_prepack(sr_ctx* ctx, const sr_pack_desc* uses, int n, void* stream);
int  sr_eltwise(sr_ctx* ctx, int op, const void* a, const void* b, float alpha, float beta, void* out, int64_t n, void* stream);
/* keras.optimizers.Adam's dense update (the optimiser of ESRGAN_model.py:176-195, SRCNN_model.py:55-60, EDSR_model.py:127-140) over one flat
 * fp32 bucket of n parameters, in place on the device: g is first multiplied by grad_scale (1 / world size after a summing all-reduce; 1 leaves
 * it alone), then m = b1 m + (1-b1) g, v = b2 v + (1-b2) g g, w -= lr_t m / (sqrt(v) + epsilon); lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) is the
 * caller's (it knows the step count).  Each operation is rounded on its own: bit for bit NumPy's fp32 evaluation of the same expression. */
int  sr_adam(sr_ctx* ctx, void* w, const void* g, void* m, void* v, int64_t n, float lr_t, float beta1, float one_minus_beta1, float beta2,
             float one_minus_beta2, float epsilon, float grad_scale, void* stream);
int  sr_space_to_depth(sr_ctx* ctx, const void* x, int B, int H, int W, int C, int r, void* y, void* stream);
/* More halves of ESRGAN._train_step's backward pass (ESRGAN_model.py:475-533), fp32 device tensors:
 * sr_matmul: C[b] = alpha * op(A[b]) op(B[b]) (row-major, op = transpose when the flag is set) -- the MATERIALISED SelfAttention of the
 *   24x24 / 48x48 training patches (:57-65) and its six backward products; sr_softmax_rows / sr_softmax_bwd: softmax over the last axis in
 *   place, and ds = p * (dp - <dp, p>).
 * sr_maxpool2_bwd: MaxPooling2D(2,2) gradient (VGG19 extractor); sr_zero_insert2: adjoint of the stride-2 pick of the discriminator's
 *   strided convs (dy [B,ceil(H/2),ceil(W/2),C] -> [B,H,W,C]); sr_spectral_l1_bwd: gradient of scale * sr_spectral_l1 w.r.t. a. */
/* sr_spatial_op: the non-conv layers of the discriminator / VGG19 graphs as single ops, x f32 [B,H,W,C] -> y:
 *   SR_SP_MAXPOOL2 [B,H/2,W/2,C] (MaxPooling2D(2,2) VALID), SR_SP_GAP [B,C] (GlobalAveragePooling2D), SR_SP_PICK2 [B,ceil(H/2),ceil(W/2),C]
 *   (the sampling half of a stride-2 SAME conv, see SR_MODEL_ESRGAN_D), SR_SP_VGG_PREPROCESS [B,H,W,3] (ESRGAN_model.py:401-408). */
enum { SR_SP_MAXPOOL2 = 0, SR_SP_GAP = 1, SR_SP_PICK2 = 2, SR_SP_VGG_PREPROCESS = 3 };
int  sr_spatial_op(sr_ctx* ctx, int op, const void* x, int B, int H, int W, int C, void* y, void* stream);
int  sr_matmul(sr_ctx* ctx, const void* A, const void* B, void* C, int batch, int M, int N, int K, int transA, int transB, float alpha, void* stream);
int  sr_softmax_rows(sr_ctx* ctx, void* s, int64_t rows, int cols, void* stream);
int  sr_softmax_bwd(sr_ctx* ctx, const void* p, const void* dp, void* ds, int64_t rows, int cols, void* stream);
int  sr_maxpool2_bwd(sr_ctx* ctx, const void* x, const void* dy, int B, int H, int W, int C, void* dx, void* stream);
int  sr_zero_insert2(sr_ctx* ctx, const void* dy, int B, int H, int W, int C, void* out, void* stream);
int  sr_spectral_l1_bwd(sr_ctx* ctx, const void* a, const void* b, int B, int H, int W, int C, float scale, void* da, void* stream);
/* add_padding + sliding-window extraction (SRCNN_model.py:127-162, EDSR_model.py:201-223,
 * ESRGAN_model.py:883-901, VGG16_model.py:216-239): img f32 [H,W,C] (unpadded), reflect padding
 * bottom/right computed from (patch,stride); out [P,patch,patch,C] of out_dtype, each value
 * v*mul+add (ESRGAN: *2-1).  *n_patches receives P.  out may be NULL to query P only. */
int  sr_extract_patches(sr_ctx* ctx, const float* img, int H, int W, int C, int patch, int stride,
                        float mul, float add, int out_dtype, void* out, int64_t out_capacity,
                        int* n_patches, void* stream);
/* reconstruct_from_patches (SRCNN_model.py:164-188, EDSR_model.py:225-256, ESRGAN_model.py:903-921):
 * overlap-add of [P,patch*scale,patch*scale,C] patches (in_dtype; value v*mul+add first, ESRGAN:
 * (v+1)/2), divide by coverage count, crop to [H*scale,W*scale,C], clip[0,1]; out f32.
 * H,W = unpadded LR size; patch,stride in LR pixels. */
int  sr_overlap_add(sr_ctx* ctx, const void* patches, int in_dtype, int H, int W, int C, int patch,
                    int stride, int scale, float mul, float add, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SR355_H */
