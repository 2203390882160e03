// ssal_internal.h -- launch wrappers shared between the kernel translation units and the C ABI.
// gfx950 (MI355X / CDNA4) only.  All tensors fp32 NHWC unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssal_measure.h"

namespace ssal {

// How the residual branch of an ENet bottleneck is merged in the epilogue of the expansion conv.
enum ResMode : int {
    RES_NONE = 0,    // no residual
    RES_ADD = 1,     // y = prelu(v + res[same pixel])                     (Bottleneck, enet_modules.py:596-598)
    RES_POOL = 2,    // y = prelu(v + maxpool2x2(res_src) zero-padded)     (Downsample, :927-937); writes 2-bit codes
    RES_UNPOOL = 3,  // y = prelu(v + unpool(res_src, code))               (Upsample,   :1285-1291)
};

struct ConvArgs {
    const float *x;      // [N,H,W,Cin]
    const float *w;      // [KH,KW,Cin,Cout]  (HWIO)
    float *y;            // [N,Ho,Wo,Cout]
    int N, H, W, Cin, Cout, KH, KW, stride, dil;
    int Ho, Wo, pad_t, pad_l;
    // epilogue: folded batch-norm (nullable) then PReLU (nullable)
    const float *scale, *shift, *alpha;
    // residual merge
    int res_mode;
    const float *res;         // RES_ADD: [N,Ho,Wo,Cout]; RES_POOL: [N,2Ho,2Wo,res_C]; RES_UNPOOL: [N,Ho/2,Wo/2,Cout]
    int res_C;
    uint8_t *code_out;        // RES_POOL: [N,Ho,Wo,res_C], code = dy*2+dx of the first maximum
    const uint8_t *code_in;   // RES_UNPOOL: [N,Ho/2,Wo/2,Cout]
    const float *res_alpha;   // PReLU after the residual add
};

hipError_t launch_conv(const ConvArgs &a, hipStream_t s);

// conv2d_transpose 3x3 stride 2 SAME.  wT is the re-laid-out kernel [3][3][Cin][Cout].
hipError_t launch_convT(const float *x, int N, int H, int W, int Cin, const float *wT, int Cout,
                        const float *scale, const float *shift, const float *alpha, float *y,
                        hipStream_t s);

// Initial block (enet_modules.py:190-224): concat[conv3x3 s2, maxpool2x2] -> BN -> PReLU, 16 channels out.
hipError_t launch_initial(const void *x, bool x_is_u8, int N, int H, int W, int Cin, const float *w,
                          const float *scale, const float *shift, const float *alpha, float *y,
                          hipStream_t s);

// Final transposed conv (enet_modules.py:1359-1381) fused with the acquisition score
// (active_learning.py:234-263).  wF is the re-laid-out kernel [3][3][16][K].
// logits (nullable) [N,2H,2W,K]; partial: [N * final_score_blocks(H,W)] doubles.
int final_score_blocks(int H, int W);
hipError_t launch_final_score(const float *x, int N, int H, int W, const float *wF, int K,
                              float *logits, int measure, float threshold, double *partial,
                              uint8_t *label, uint8_t *mask, float *conf, hipStream_t s);
// Bottleneck5_1 evaluated inside the Final + score kernel (score-only form; x5 = Bottleneck5_0's output [N,H,W,16])
hipError_t launch_bnk4_final_score(const float *x5, int N, int H, int W, const float *wp, const float *ps, const float *pt,
                                   const float *pa, const float *wc, const float *cs, const float *ct, const float *ca,
                                   const float *we, const float *es, const float *et, const float *ra, const float *wF, int K,
                                   int measure, double *partial, hipStream_t s);
// scores[n] = sum(partial[n, 0..blocks)) / pixels, fixed summation order (bitwise reproducible)
hipError_t launch_reduce_mean(const double *partial, int N, int blocks, double pixels,
                              double *scores, hipStream_t s);

// stand-alone score on materialised logits [N,H,W,K]
int score_blocks(int H, int W);
hipError_t launch_score_logits(const float *logits, int N, int H, int W, int K, int measure,
                               float threshold, double *partial, uint8_t *label, uint8_t *mask,
                               float *conf, hipStream_t s);

// masked softmax cross-entropy forward (tensortools/losses.py:3-74); partial: 2 * xent_blocks doubles
int xent_blocks(int H, int W);
hipError_t launch_masked_xent(const float *logits, const uint8_t *labels, const float *mask, int N,
                              int H, int W, int K, float weight, float label_smoothing,
                              double *partial, double *out, hipStream_t s);

// pooling / unpooling with reference int64 indices
hipError_t launch_maxpool_argmax(const float *x, int N, int H, int W, int C, float *y,
                                 int64_t *argmax, int include_batch, hipStream_t s);
hipError_t launch_unpool_scatter(const float *x, const int64_t *idx, int N, int H, int W, int C,
                                 int idx_has_batch, float *y, hipStream_t s);
hipError_t launch_argmax_to_codes(const int64_t *argmax, int N, int Ho, int Wo, int C, uint8_t *code,
                                  int *bad, hipStream_t s);
hipError_t launch_codes_to_argmax(const uint8_t *code, int N, int Ho, int Wo, int C,
                                  int64_t *argmax, hipStream_t s);

// MFMA-fused regular / dilated bottleneck (ssal_bottleneck_mfma.hip)
bool bottleneck_mfma_supported(int Cin, int f, bool asym);
hipError_t launch_bottleneck_mfma(const float *x, float *y, int N, int H, int W, int Cin, int dil,
                                  const float *wp, const float *ps, const float *pt, const float *pa,
                                  const float *wc, const float *wc2 /* asym: (1,5) kernel, else NULL */,
                                  const float *cs, const float *ct, const float *ca, const float *we,
                                  const float *es, const float *et, const float *ra, hipStream_t s,
                                  const float *wq = nullptr /* 128-channel non-asymmetric blocks: wp | wc | we in quad layout (ssal_host.h) */);
// opt-in SSAL_ARITH_BF16X3 form of the 128-channel regular / dilated / asymmetric block (ssal_bottleneck_bf16x3.hip);
// packed = the layer's kernels in ssal_bf16x3.h layout (a.wc2 != NULL selects the asymmetric kernel)
struct BnkArgs;
bool bottleneck_bf16x3_supported(int Cin, int f);
hipError_t launch_bottleneck_bf16x3(const BnkArgs &a, const void *packed, hipStream_t s);
struct DownArgs;
bool downsample_bf16x3_supported(int Cin, int Cout);
hipError_t launch_downsample_bf16x3(const DownArgs &a, const void *packed, hipStream_t s);
struct UpArgs;
bool upsample_bf16x3_supported(int Cin, int Cout);
hipError_t launch_upsample_bf16x3(const UpArgs &a, const void *packed, hipStream_t s);
// MFMA-fused downsample bottleneck 64 -> 128 / 16 -> 64 (writes the 2x2 window codes)
bool downsample_mfma_supported(int Cin, int Cout);
hipError_t launch_downsample_mfma(const float *x, float *y, uint8_t *code, int N, int H, int W,
                                  int Cin, const float *wp, const float *ps, const float *pt, const float *pa,
                                  const float *wc, const float *cs, const float *ct, const float *ca,
                                  const float *we, const float *es, const float *et, const float *ra,
                                  hipStream_t s);
// Initial block + Bottleneck1_0 in one launch (ssal_bottleneck_mfma16.hip: k_initial_down16); H, W = image dims
bool initial_down16_supported(int c_in);
hipError_t launch_initial_down16(const void *img, bool img_is_u8, int N, int H, int W, int c_in, const float *iw,
                                 const float *iscale, const float *ishift, const float *ialpha, float *y, uint8_t *code,
                                 const float *wp, const float *ps, const float *pt, const float *pa, const float *wc,
                                 const float *cs, const float *ct, const float *ca, const float *we, const float *es,
                                 const float *et, const float *ra, hipStream_t s);
// MFMA-fused upsample bottleneck 128 -> 64 / 64 -> 16 (window-code unpooling); ws = stacked transposed-conv kernel
bool upsample_mfma_supported(int Cin, int Cout);
hipError_t launch_upsample_mfma(const float *x, float *y, const uint8_t *code, int N, int H, int W,
                                int Cin, const float *wp, const float *ps, const float *pt, const float *pa,
                                const float *ws, const float *cs, const float *ct, const float *ca,
                                const float *we, const float *es, const float *et, const float *wr,
                                const float *ra, hipStream_t s);
// Tuning knobs of the fused bottleneck launchers (defaults = the shipping configuration; changed at run time with
// ssal_debug_set_knob for A/B runs and for the tests that compare the variants bit for bit: every setting of the
// product build produces identical results.  Work-skipping "ablate" and the SSAL_* environment reads exist only in
// -DSSAL_MEASURE builds).
constexpr int IC_FRONT_DEFAULT = 3, IC_DUAL_DEFAULT = 1, ASYM_TW16_DEFAULT = 1, IG_SB_DEFAULT = 3, IC_GROUPS_DEFAULT = 1, BNK_QEPI_DEFAULT = 2;
struct Knobs {
    int bnk_tw;      // 16 = force 8x16 tiles in the 128-channel bottleneck kernels
    int bnk_o4;      // k_bottleneck_o4 (8x16 tiles, four workgroups per CU): 2 (default) = where the phase sub-image is <= 16 wide, 1 = everywhere, 0 = never
    int bnk_xcd;     // 1 = XCD-aware tile order in the 128-channel bottleneck kernels
    int bnk_qepi;    // k_bottleneck_mfma<32> epilogue: 2 (default) = D[co][pixel], 16-byte residual loads, whole-row 16-byte stores through quad_transpose4; 1 = the same without the transpose; 0 = D[pixel][co] with 4-byte accesses (rounds 1-4)
    int asym_tw16;   // asymmetric 128-channel block: 1 = k_bottleneck_mfma_asym16x (8x16 tiles, 3 workgroups per CU, (5,1) result written over the projected rows), 0 = the 8x32 / two-halves kernel
    int fuse_ends;   // bit 0: Initial + Bottleneck1_0 in one launch; bit 1: Bottleneck5_1 inside Final + score (ranking pass); default 3
    int ic_groups;   // ICNet score path: image-group chains (1, default: with the three-workgroup up-sampling kernel one chain is 2.5 % faster than two; ENet keeps img_groups = 2)
    int ig_sb;       // k_igemm with ONE LDS buffer at three workgroups per CU: 0 = never, 1 = the up-sampling form (UP2 = 2), 2 = that and the plain form at NT = 2, 3 (default) = also at NT = 1
    int ic_dual;     // ICNet score path: 1 = a block's projection shortcut is evaluated inside its increase launch (k_igemm<.., DUAL>)
    int ic_front;    // ICNet score path: bit 0 = conv1_sub1 + conv2_sub1 in one launch (k_front2<s2>), bit 1 = conv1_1_3x3_s2 + conv1_2_3x3 (k_front2<s1>)
    int ig_div;      // ICNet: the ">= 512 workgroups per launch" rules of k_igemm / k_conv3x3_c32 use 512 / ig_div; 0 (default) = the number of image-group chains of the call
    int img_lag;     // chain g of the image-group schedule starts this many layers behind chain g - 1 (default 0)
    int img_span;    // which layers run in image groups: 0 = Bottleneck2_1..3_8, 1 = + 2_0, 2 = 1_0..5_1, 3 = Initial..5_1, 4 = Initial..Final + score (default)
    int img_groups;  // ENet: the layers of img_span run as this many image groups on side streams (default 2; 1 = everything on the caller's stream)
#ifdef SSAL_MEASURE
    int ablate;      // measurement builds only: 1 = stop after the projection phase, 2 = skip it (results invalid)
    int bnk_split;   // measurement builds only: 1 = the regular 128-channel bottleneck on bf16x3 split operands (ssal_split_probe.hip; NOT bit-identical)
#endif
};
Knobs &knobs();
bool mfma_family();  // ssal_set_kernel_family: true = the MFMA-fused kernels (default), false = generic kernels everywhere
// Image-group schedule plumbing shared by the ENet and ICNet handles.  Side streams: ONE pool per DEVICE, shared by every
// handle of the process (created on first use, never destroyed) -- per-handle streams would exhaust the hardware queues of
// the process (with more user streams than queues two chains share a queue and the second model scored in a process ran
// 14 % slower).  Events: private to each CALL, drawn from a mutex-protected per-device pool and handed back once the join
// has been enqueued (a wait that is already enqueued keeps the record it saw; re-recording the event later is harmless),
// so any number of host threads may drive one handle on their own streams + workspaces.
hipError_t side_stream(int g, hipStream_t *out);  // pool of the CURRENT device
struct ChainSet {
    int G = 0;
    bool open = false;
    hipStream_t caller = nullptr, side[8] = {};
    hipEvent_t fork_ev = nullptr, join_ev[8] = {};
    // records the fork point on `s` and makes the G side streams of the current device wait for it
    hipError_t begin(int groups, hipStream_t s);
    // chain g has reached a point chain g2 must wait for (img_lag experiments)
    hipError_t link(int g, int g2);
    // joins every chain into the caller's stream and returns the events; also the error path: a call that fails half way
    // still joins, so that no chain keeps using the caller's workspace after the call has returned
    hipError_t end();
    ~ChainSet() { if (open) (void)end(); }
};

// phase-trace buffer the bottleneck launchers hand to their kernels (ssal_debug_set_trace; NULL = off)
extern unsigned long long *g_trace_buf;
extern long g_trace_bytes;
hipError_t launch_probe_swap(float *out, hipStream_t s);
#ifdef SSAL_MEASURE
struct BnkArgs;
hipError_t launch_bottleneck_split(const BnkArgs &a, hipStream_t s);  // ssal_split_probe.hip (measurement libraries only)
#endif
hipError_t launch_copy_probe(int mode, const float *x, float *y, int N, int H, int W, int spin, hipStream_t s);
hipError_t launch_mfma_peak(int shape, int blocks, int iters, float *out, hipStream_t s);

hipError_t launch_prelu(const float *x, int64_t pixels, int C, const float *alpha, float *y,
                        hipStream_t s);
hipError_t launch_affine(const float *x, int64_t pixels, int C, const float *scale,
                         const float *shift, float *y, hipStream_t s);
hipError_t launch_bn_fold(const float *mean, const float *var, const float *gamma,
                          const float *beta, int C, float *scale, float *shift, hipStream_t s);
// xops.spatial_dropout (extra_ops.py:137-151): y = (x / (1-rate)) * floor(1-rate + u(n,c)), u = hash(seed, n*C + c)
hipError_t launch_spatial_dropout(const float *x, int N, int64_t pixels_per_image, int C, float rate, uint64_t seed,
                                  float *y, hipStream_t s);
hipError_t launch_resize_bilinear(const float *x, int N, int H, int W, int C, int OH, int OW,
                                  float *y, hipStream_t s);
hipError_t launch_synth_frames(uint64_t seed, int64_t first, int count, int H, int W, int C,
                               void *out, bool out_is_u8, hipStream_t s);

}  // namespace ssal
