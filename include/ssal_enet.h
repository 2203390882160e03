/*
 * ssal_enet.h -- C ABI of libssal_hip.so: the MI355X (gfx950) pool-scoring hot path.
 *
 * The reference (alfrunesiq/SemanticSegmentationActiveLearning) has no FFI: the path sits behind a
 * Python object API (models.ENet, xops.*, and score tensors assembled inline in active_learning.py)
 * executed by the TensorFlow runtime.  Each entry point below states the reference interface it
 * replaces (file:line, relative to the reference root).  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types (a stream is passed as void* = hipStream_t).
 *   - every function returns an int status (SSAL_OK == 0); nothing throws across the boundary.
 *     ssal_last_error() returns a thread-local message for the last non-zero status.
 *   - "dev" pointers are device (HBM) pointers owned by the caller and only borrowed for the call;
 *     "host" pointers are host memory.  The library owns only the weights inside a handle.
 *   - activations are fp32 NHWC, conv kernels HWIO, transposed-conv kernels HW-O-I (TF layouts);
 *     all launches are stream-ordered and asynchronous.
 *   - threading: forward / score / run_layer on a COMMITTED handle may be called from any number of host threads at
 *     the same time, provided every concurrent call has its own stream and its own workspace (the image-group schedule
 *     draws its fork / join events per call from a mutex-protected pool; its side streams are one process-wide pool per
 *     device).  set_tensor / commit / destroy must not overlap any other call on the same handle.  A handle lives on
 *     ONE device -- the device that was current at commit; a call made with another current device returns
 *     SSAL_ESTATE (create one handle per device).  A call that fails half way still joins its side-stream chains into
 *     the caller's stream before it returns.
 */
#ifndef SSAL_ENET_H
#define SSAL_ENET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAL_OK        0
#define SSAL_EINVAL    1 /* bad argument (maps to ValueError) */
#define SSAL_EHIP      2 /* HIP runtime error */
#define SSAL_ENOTIMPL  3 /* maps to NotImplementedError (active_learning.py:259-260) */
#define SSAL_ESTATE    4 /* handle not committed / tensor missing */
#define SSAL_ENOMEM    5 /* workspace too small */

/* acquisition measures, active_learning.py:239-260 (conf/default_params.json "measure") */
#define SSAL_MEASURE_ENTROPY     0
#define SSAL_MEASURE_MARGIN      1
#define SSAL_MEASURE_CONFIDENCE  2

/* arithmetic modes of the *_arith entry points.
 *   SSAL_ARITH_F32     the default and the only mode of every other entry point: exact fp32 (fmaf chains in (kh, kw, ci)
 *                      order on v_mfma_f32_*), bit-identical to the parity oracle.
 *   SSAL_ARITH_BF16X3  OPT-IN: the 128-channel regular / dilated / asymmetric bottlenecks (Bottleneck2_1 .. 3_8), the
 *                      downsample block Bottleneck2_0 and the upsample block Bottleneck4_0 (18 of the 29 launches) evaluate their convolutions on v_mfma_f32_32x32x16_bf16 with every fp32 operand
 *                      split into three bf16 terms and the six leading cross products accumulated in fp32 (what is dropped
 *                      is O(2^-24) relative per product).  Same accuracy class as fp32, a different summation: logits differ
 *                      from the default mode by <= ~1e-5, per-pixel confidences by <= 1e-4 (north_star's tolerance), per-image
 *                      scores by <= 1e-6; the max-pool + argmax of both pooling blocks is evaluated in exact fp32 (inside
 *                      Bottleneck2_0's split-operand kernel too), so the pooling indices are bit-identical.  No reference counterpart (the reference computes in fp32 on TensorFlow). */
#define SSAL_ARITH_F32     0
#define SSAL_ARITH_BF16X3  1

typedef struct ssal_enet ssal_enet;

const char *ssal_version(void);
const char *ssal_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Model handle: replaces the models.ENet object (models/__init__.py:1-3, models/enet/enet.py:6-407).
 * Tensor names are the reference attribute names "<Layer>.<attr>", e.g. "Initial.kernel",
 * "Bottleneck2_3.conv_kernel.0" (KernelCol, [5,1,f,f]) / ".1" (KernelRow, [1,5,f,f]),
 * "Bottleneck4_0.res_kernel", "Final.kernel"  (enet_modules.py:139-187,366-523,730-865,1070-1214,1349-1356).
 * ---------------------------------------------------------------------------------------------- */
int ssal_enet_create(int c_in, int classes, ssal_enet **out);
int ssal_enet_destroy(ssal_enet *net);
int ssal_enet_num_tensors(const ssal_enet *net);
int ssal_enet_tensor_info(const ssal_enet *net, int i, const char **name, int *ndim, int64_t dims[4]);
/* copy one parameter tensor from host memory into the handle (staged until commit) */
int ssal_enet_set_tensor(ssal_enet *net, const char *name, const float *host, int64_t numel);
/* fold the batch-norm statistics (extra_ops.py:181-184, eps=1e-3), re-layout kernels and upload */
int ssal_enet_commit(ssal_enet *net, void *stream);
/* bytes of device scratch needed by forward/score for a batch of n images of h x w */
int64_t ssal_enet_workspace_bytes(const ssal_enet *net, int n, int h, int w);

/* ENet.call(inputs, training=False) -> logits   (models/enet/enet.py:320-407)
 * x_dev: [n,h,w,c_in] fp32, logits_dev: [n,h,w,classes] fp32; h,w divisible by 8. */
int ssal_enet_forward_nhwc(ssal_enet *net, const float *x_dev, int n, int h, int w,
                           float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream);

/* ENet.call + softmax + acquisition measure + float64 per-image mean, fused
 * (active_learning.py:229-263: pseudo_logits, pseudo_label, pseudo_prob, pseudo_confidence,
 *  pseudo_mean_confidence, pseudo_mask).  The logits never reach HBM.
 * scores_dev: [n] float64 (required).  Optional outputs (NULL to skip):
 *   label_dev [n,h,w] uint8  = argmax_k logits                      (:234-236)
 *   mask_dev  [n,h,w] uint8  = conf < threshold ? 0 : 1             (:265-269)
 *   conf_dev  [n,h,w] fp32   = per-pixel confidence                 (:243-258) */
int ssal_enet_score_nhwc(ssal_enet *net, const float *x_dev, int n, int h, int w, int measure,
                         float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                         float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream);

/* The same two entry points on the DECODED frame: x_dev [n,h,w,c_in] uint8.  The reference converts right after
 * decoding, `tf.image.convert_image_dtype(image, tf.float32)` = u8 * float32(1/255) (tensortools/input.py:289-290);
 * here the Initial block does that conversion on the fly: identical bits out, a quarter of the bytes over PCIe
 * and into the first kernel. */
int ssal_enet_forward_nhwc_u8(ssal_enet *net, const uint8_t *x_dev, int n, int h, int w,
                              float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream);
int ssal_enet_score_nhwc_u8(ssal_enet *net, const uint8_t *x_dev, int n, int h, int w, int measure,
                            float threshold, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                            float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream);

/* forward / score with an explicit arithmetic mode (SSAL_ARITH_*); x_dev is float32 (x_is_u8 == 0) or the decoded uint8
 * frame (x_is_u8 != 0).  arithmetic == SSAL_ARITH_F32 is exactly ssal_enet_forward_nhwc / ssal_enet_score_nhwc (_u8).
 * Replaces the same reference code (models/enet/enet.py:320-407, active_learning.py:229-263). */
int ssal_enet_forward_nhwc_arith(ssal_enet *net, const void *x_dev, int x_is_u8, int n, int h, int w, int arithmetic,
                                 float *logits_dev, void *ws_dev, int64_t ws_bytes, void *stream);
int ssal_enet_score_nhwc_arith(ssal_enet *net, const void *x_dev, int x_is_u8, int n, int h, int w, int measure,
                               float threshold, int arithmetic, double *scores_dev, uint8_t *label_dev, uint8_t *mask_dev,
                               float *conf_dev, void *ws_dev, int64_t ws_bytes, void *stream);

/* Byte offsets into the workspace of the last forward/score call of the tensors behind
 * ENet.endpoint_outputs (models/enet/enet.py:311-318): offs[0] bottleneck5_1 [n,h/2,w/2,16],
 * offs[1] bottleneck4_2 [n,h/4,w/4,64], offs[2] bottleneck3_8 [n,h/8,w/8,128]. */
int ssal_enet_endpoint_offsets(const ssal_enet *net, int n, int h, int w, int64_t offs[3]);

/* The max-pooling indices ENet.call hands from Bottleneck1_0 / Bottleneck2_0 to Bottleneck5_0 / Bottleneck4_0
 * (models/enet/enet.py:331,338,359,364), of the LAST forward/score call that ran on this workspace, converted from
 * the internal 1-byte window codes to the reference's int64 per-image index (y*W + x)*C + c:
 * which = 1 -> argmax1 [n,h/4,w/4,16], which = 2 -> argmax2 [n,h/8,w/8,64]. */
int ssal_enet_export_argmax(const ssal_enet *net, const void *ws_dev, int64_t ws_bytes, int n, int h, int w,
                            int which, int64_t *argmax_out_dev, void *stream);

/* Run ONE layer of the handle (Layer.__call__ of enet_modules.py: Initial :190-224, Bottleneck
 * :526-599, BottleneckDownsample :868-938, BottleneckUpsample :1217-1292, Final :1359-1381).
 * x_dev [n,h,w,cin] -> y_dev (shape by layer kind).  argmax tensors use the reference's int64
 * per-image index (y*W + x)*C + c: argmax_out_dev is written by a Downsample
 * layer, argmax_in_dev is consumed by an Upsample layer; NULL otherwise. */
int ssal_enet_run_layer(ssal_enet *net, const char *layer, const float *x_dev, int n, int h, int w,
                        float *y_dev, int64_t *argmax_out_dev, const int64_t *argmax_in_dev,
                        void *ws_dev, int64_t ws_bytes, void *stream);
int64_t ssal_enet_layer_workspace_bytes(const ssal_enet *net, const char *layer, int n, int h, int w);
/* the same with an arithmetic mode (layers the mode has no kernel for run in exact fp32) */
int ssal_enet_run_layer_arith(ssal_enet *net, const char *layer, const float *x_dev, int n, int h, int w, int arithmetic,
                              float *y_dev, int64_t *argmax_out_dev, const int64_t *argmax_in_dev, void *ws_dev,
                              int64_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone operators
 * ---------------------------------------------------------------------------------------------- */

/* softmax + measure + float64 mean on materialised logits (active_learning.py:239-263) */
int64_t ssal_score_workspace_bytes(int n, int h, int w);
int ssal_score_logits_nhwc(const float *logits_dev, int n, int h, int w, int classes, int measure,
                           float threshold, double *scores_dev, uint8_t *label_dev,
                           uint8_t *mask_dev, float *conf_dev, void *ws_dev, int64_t ws_bytes,
                           void *stream);

/* tensortools.losses.masked_softmax_cross_entropy forward (tensortools/losses.py:3-74): label
 * smoothing, optional ENet-style class weighting (weight > 1), fp32 sum over the batch axis, float64
 * over the spatial axes, divided by the (fp32) mask sum.  labels uint8 [n,h,w], mask fp32 [n,h,w],
 * loss_dev: one float64. */
int64_t ssal_xent_workspace_bytes(int h, int w);
int ssal_masked_softmax_cross_entropy(const float *logits_dev, const uint8_t *labels_dev,
                                      const float *mask_dev, int n, int h, int w, int classes,
                                      float weight, float label_smoothing, double *loss_dev,
                                      void *ws_dev, int64_t ws_bytes, void *stream);

/* tf.nn.max_pool_with_argmax(ksize 2x2, strides 2, SAME, Targmax=int64) (enet_modules.py:927-929);
 * include_batch selects the TF<=1.13 CPU index convention (extra_ops.py:63-81). */
int ssal_max_pool_with_argmax_2x2(const float *x_dev, int n, int h, int w, int c, float *y_dev,
                                  int64_t *argmax_dev, int include_batch, void *stream);
/* xops.unpool_2d(inputs, idx, strides=[1,2,2,1])  (models/util/extra_ops.py:28-86).  Indices must be unique per
 * output element (true of pooling-derived indices, the reference's only use): tf.scatter_nd sums duplicates, this
 * scatter assigns. */
int ssal_unpool_2d(const float *x_dev, const int64_t *idx_dev, int n, int h, int w, int c,
                   int idx_has_batch, float *y_dev, void *stream);
/* xops.prelu(x, alpha)  (models/util/extra_ops.py:9-26) */
int ssal_prelu(const float *x_dev, int64_t pixels, int c, const float *alpha_dev, float *y_dev,
               void *stream);
/* xops.spatial_dropout(inputs, drop_rate)  (models/util/extra_ops.py:137-151; called, training only, at
 * enet_modules.py:591-594): tf.nn.dropout with noise_shape [N,1,1,C]: y = (x / (1-rate)) * floor((1-rate) + u[n,c]),
 * one uniform draw per (image, channel) plane.  u is a counter-based hash of (seed, n*C + c) -- TensorFlow's random
 * stream is not reproducible, the distribution and the arithmetic are.  x_dev/y_dev [n, pixels_per_image, c]. */
int ssal_spatial_dropout(const float *x_dev, int n, int64_t pixels_per_image, int c, float rate, uint64_t seed,
                         float *y_dev, void *stream);
/* xops.batch_norm(..., training=False)  (models/util/extra_ops.py:154-185) */
int ssal_batch_norm_inference(const float *x_dev, int64_t pixels, int c, const float *mean_dev,
                              const float *var_dev, const float *gamma_dev, const float *beta_dev,
                              float *y_dev, void *stream);
/* tf.nn.conv2d(x, kernel HWIO, strides [1,s,s,1], dilations [1,d,d,1], "SAME")
 * (enet_modules.py:205,538,554,559,565,581,880,895,911,1236,1267,1285) */
int ssal_conv2d_same(const float *x_dev, int n, int h, int w, int cin, const float *kernel_dev,
                     int kh, int kw, int cout, int stride, int dilation, float *y_dev, void *stream);
/* tf.nn.conv2d_transpose(x, kernel [3,3,cout,cin], strides 2, "SAME") -> [n,2h,2w,cout]
 * (enet_modules.py:1251-1255, 1376-1380) */
int ssal_conv2d_transpose_3x3_s2(const float *x_dev, int n, int h, int w, int cin,
                                 const float *kernel_dev, int cout, float *y_dev, void *stream);
/* tf.image.resize_bilinear(x, [oh,ow]) with TF-1.13 defaults (align_corners=False, legacy
 * src = dst * in/out mapping)  (inference.py:96-99) */
int ssal_resize_bilinear(const float *x_dev, int n, int h, int w, int c, int oh, int ow,
                         float *y_dev, void *stream);

/* Synthetic Cityscapes-shaped frames for benchmarking/tests: frame f of the pool is a
 * pure function of (seed, f); out_dev [count,h,w,c] fp32 = uint8 pixel * (1/255)
 * (tensortools/input.py:289-290 convert_image_dtype).  Host twin: synthetic.synth_frames_u8(). */
int ssal_synth_frames_nhwc(uint64_t seed, int64_t first_frame, int count, int h, int w, int c,
                           float *out_dev, void *stream);
/* the same frames before the conversion: out_dev [count,h,w,c] uint8 */
int ssal_synth_frames_nhwc_u8(uint64_t seed, int64_t first_frame, int count, int h, int w, int c,
                              uint8_t *out_dev, void *stream);

/* The switches below (ssal_set_kernel_family, ssal_debug_set_knob, ssal_debug_set_trace, ssal_profile_*) are
 * PROCESS-GLOBAL measurement aids, not part of the re-entrant per-handle / per-stream contract stated at the top of
 * this header: set them from one host thread while no other thread is inside the library.
 *
 * Kernel-family switch for A/B measurements and cross-checks (no reference counterpart):
 * 1 = MFMA-fused bottleneck kernels on the shapes they support (default), 0 = generic kernels
 * everywhere.  Both families produce bit-identical results. */
int ssal_set_kernel_family(int use_mfma);
/* hardware-assumption probe used by the tests (v_permlane32_swap / v_permlane16_swap lane
 * semantics): writes 256 floats */
int ssal_debug_probe(float *out_dev_256, void *stream);

/* tuning / A-B knob of the fused bottleneck launchers ("bnk_tw": 16 forces 8x16 tiles, "bnk_o4": the four-workgroups-per-CU
 * form of the regular 128-channel block -- 2 (default) where the phase sub-image is at most 16 pixels wide, 1 everywhere,
 * 0 never --, "bnk_xcd": 0 switches the
 * XCD-aware tile order off, "img_groups": G runs the layers selected by "img_span" (default 4 = Initial .. Final + score) as G image
 * groups on G library-owned side streams, forked from / joined into the caller's stream with events -- default 2, 1 =
 * everything on the caller's stream; "ic_front": ICNet score path, bit 0 (default 1) = conv1_sub1 + conv2_sub1 as one launch,
 * bit 1 (default 0) = conv1_1_3x3_s2 + conv1_2_3x3 as one launch; "ic_dual": ICNet score path, 1 (default) = a block's projection
 * shortcut is evaluated inside its 1x1 increase launch).  Every setting produces bit-identical results
 * (tests/test_gpu_parity.py, tests/test_icnet_gpu.py); SSAL_EINVAL for
 * an unknown name.  The product build reads no environment variable and contains no work-skipping switch: phase
 * ablation ("ablate") and the SSAL_* environment defaults exist only in -DSSAL_MEASURE builds (tools/phase_trace.py),
 * whose ssal_version() says so. */
int ssal_debug_set_knob(const char *name, int value);
/* JSON object with the state of every switch that can change what a launch does or costs: kernel_family, bnk_tw, bnk_o4,
 * bnk_xcd, img_groups, img_span, fuse_ends, img_lag, ig_div, ic_front, ic_dual, ablate, measure_build, profiling, defaults (1 iff all are at their shipping values).  bench.py prints it
 * in its result line and refuses to time anything else. */
int ssal_debug_get_knobs(char *json_out, int64_t cap);

/* measurement aid, only functional in a -DSSAL_PHASE_TRACE build (tools/phase_trace.py; SSAL_ENOTIMPL
 * otherwise): the fused bottleneck kernels write 16 x uint64 per wave (shader-clock phase marks,
 * 100 MHz realtime of first / last mark, HW_ID, XCC_ID) into buf_dev; NULL switches it off. */
int ssal_debug_set_trace(void *buf_dev, int64_t bytes);

/* Measurement aid (no reference counterpart): when enabled, every kernel launch is bracketed by
 * HIP events on its own stream; ssal_profile_collect() returns per-kernel launch counts, total
 * milliseconds and ALGORITHMIC flops / bytes as a JSON object.  Single host thread only. */
int ssal_profile_enable(int on);
int ssal_profile_collect(char *json_out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SSAL_ENET_H */
