// ssal_kernels.hip -- generic (shape-agnostic) HIP kernels of the ENet pool-scoring path, gfx950.
//
// These are the correctness-first kernels: every conv output element is one fp32 fmaf chain over
// (kh, kw, ci) in ascending order, exactly the order the parity oracle uses, so results are
// bit-comparable.  Lanes run along the output-channel axis so weight loads and activation stores are
// coalesced 128-B segments and the activation operand is a wave-broadcast float4.  The MFMA-tiled
// bottleneck kernels (ssal_bottleneck_mfma.hip) replace these on the hot shapes; these remain the
// path for the odd shapes (4-channel input, 4/8-wide bottlenecks) and the per-operator C ABI.
//
// Reference semantics restated per kernel (file:line relative to the reference repository).
#include "ssal_internal.h"
#include "ssal_prof.h"
#include "ssal_score.h"
#include <float.h>

namespace ssal {

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float prelu_f(float v, float a) { return v >= 0.0f ? v : a * v; }

// ------------------------------------------------------------------------------------------------
// Initial block: concat[ conv3x3 s2 SAME (Cin -> 16-Cin), maxpool2x2 s2 (Cin) ] -> BN(16) -> PReLU(16)
// (enet_modules.py:190-224; concat order conv first :214-215).  SAME on even H,W: pad (0 before, 1 after).
// One thread per output pixel; kernel weights are wave-uniform (scalar loads).
// TX = float: the reference's tensor (train_image_raw, float32 in [0,1]); TX = uint8_t: the decoded frame,
// converted on the fly exactly as tf.image.convert_image_dtype does (input.py:289-290): x = u8 * f32(1/255)
// -- a quarter of the bytes over PCIe and into this kernel, identical bits out.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_unit(float v) { return v; }
__device__ __forceinline__ float to_unit(uint8_t v) { return (float)v * (1.0f / 255.0f); }

template <int CIN, typename TX>
__global__ __launch_bounds__(256) void k_initial(const TX *__restrict__ x,
                                                 const float *__restrict__ w,
                                                 const float *__restrict__ scale,
                                                 const float *__restrict__ shift,
                                                 const float *__restrict__ alpha,
                                                 float *__restrict__ y, int N, int H, int W)
{
    constexpr int CC = 16 - CIN;
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
         p += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % Wo);
        const int oy = (int)((p / Wo) % Ho);
        const int n = (int)(p / ((long)Wo * Ho));
        float acc[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = 0.0f;
        float best[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) best[c] = -FLT_MAX;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = 2 * oy + kh;
            if (iy >= H) continue;
            // the 3 x CIN values of this kernel row are contiguous in memory: fetch them with the widest loads
            // their alignment allows (float input: element offset 2*ox*CIN is even -> 8-byte aligned)
            const TX *xrow = x + (((long)n * H + iy) * W + 2 * ox) * CIN;
            const bool third = 2 * ox + 2 < W;  // SAME padding on even W: the last column has no third pixel
            float rowv[3 * CIN];
            if (sizeof(TX) == 4 && (2 * CIN) % 2 == 0) {
                const float2 *x2 = reinterpret_cast<const float2 *>(xrow);
#pragma unroll
                for (int q = 0; q < CIN; ++q) {  // pixels 0 and 1: 2*CIN floats
                    const float2 v = x2[q];
                    rowv[2 * q] = v.x; rowv[2 * q + 1] = v.y;
                }
#pragma unroll
                for (int q = 0; q < CIN; ++q) rowv[2 * CIN + q] = third ? to_unit(xrow[2 * CIN + q]) : 0.0f;
            } else {
#pragma unroll
                for (int q = 0; q < 3 * CIN; ++q) rowv[q] = (q < 2 * CIN || third) ? to_unit(xrow[q]) : 0.0f;
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (kw == 2 && !third) continue;
                float xv[CIN];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) xv[ci] = rowv[kw * CIN + ci];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float *wr = w + ((kh * 3 + kw) * CIN + ci) * CC;
#pragma unroll
                    for (int co = 0; co < CC; ++co) acc[co] = fmaf(xv[ci], wr[co], acc[co]);
                }
                if (kh < 2 && kw < 2) {  // 2x2 pooling window, (y,x) order, strict '>'
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
                        if (xv[ci] > best[ci]) best[ci] = xv[ci];
                }
            }
        }
        float out[16];
#pragma unroll
        for (int c = 0; c < CC; ++c) out[c] = acc[c];
#pragma unroll
        for (int c = 0; c < CIN; ++c) out[CC + c] = best[c];
#pragma unroll
        for (int c = 0; c < 16; ++c) out[c] = prelu_f(fmaf(out[c], scale[c], shift[c]), alpha[c]);
        float4 *yp = reinterpret_cast<float4 *>(y + p * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            yp[q] = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    }
}

template <typename TX>
static hipError_t launch_initial_t(const TX *x, int N, int H, int W, int Cin, const float *w, const float *scale,
                                   const float *shift, const float *alpha, float *y, hipStream_t s)
{
    const long total = (long)N * (H / 2) * (W / 2);
    const int grid = cdiv(total, 256);
    ProfScope prof("k_initial", 2.0 * total * 9 * Cin * (16 - Cin),
                   (double)sizeof(TX) * N * H * W * Cin + 4.0 * total * 16.0, s);
    if (Cin == 3)
        hipLaunchKernelGGL((k_initial<3, TX>), dim3(grid), dim3(256), 0, s, x, w, scale, shift, alpha, y, N, H, W);
    else if (Cin == 4)
        hipLaunchKernelGGL((k_initial<4, TX>), dim3(grid), dim3(256), 0, s, x, w, scale, shift, alpha, y, N, H, W);
    else if (Cin == 1)
        hipLaunchKernelGGL((k_initial<1, TX>), dim3(grid), dim3(256), 0, s, x, w, scale, shift, alpha, y, N, H, W);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_initial(const void *x, bool x_is_u8, int N, int H, int W, int Cin, const float *w,
                          const float *scale, const float *shift, const float *alpha, float *y,
                          hipStream_t s)
{
    return x_is_u8 ? launch_initial_t((const uint8_t *)x, N, H, W, Cin, w, scale, shift, alpha, y, s)
                   : launch_initial_t((const float *)x, N, H, W, Cin, w, scale, shift, alpha, y, s);
}

// ------------------------------------------------------------------------------------------------
// Generic tf.nn.conv2d "SAME" (cross-correlation, NHWC x HWIO) with fused epilogue:
//   v = acc; [v = fmaf(v, scale, shift)]; [v = prelu(v, alpha)];
//   residual merge (ResMode); [v = prelu(v, res_alpha)]
// thread <-> (pixel group of PX consecutive ox, output channel); lanes run along co.
// Requires Cin % 4 == 0 and 256 % Cout == 0.
// ------------------------------------------------------------------------------------------------
template <int PX>
__global__ __launch_bounds__(256) void k_conv(ConvArgs a)
{
    const int tid = threadIdx.x;
    const int co = tid % a.Cout;
    const int gpb = 256 / a.Cout;
    const int gx = (a.Wo + PX - 1) / PX;
    const long ngroups = (long)a.N * a.Ho * gx;
    const long g = (long)blockIdx.x * gpb + tid / a.Cout;
    if (g >= ngroups) return;
    const int gxi = (int)(g % gx);
    const int oy = (int)((g / gx) % a.Ho);
    const int n = (int)(g / ((long)gx * a.Ho));
    const int ox0 = gxi * PX;

    float acc[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) acc[i] = 0.0f;

    for (int kh = 0; kh < a.KH; ++kh) {
        const int iy = oy * a.stride - a.pad_t + kh * a.dil;
        if (iy < 0 || iy >= a.H) continue;
        const float *xrow = a.x + ((long)n * a.H + iy) * a.W * a.Cin;
        for (int kw = 0; kw < a.KW; ++kw) {
            const int ixb = ox0 * a.stride - a.pad_l + kw * a.dil;
            const float *wp = a.w + (long)((kh * a.KW + kw) * a.Cin) * a.Cout + co;
            bool ok[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const int ix = ixb + i * a.stride;
                ok[i] = (ix >= 0) && (ix < a.W) && (ox0 + i < a.Wo);
            }
            for (int ci = 0; ci < a.Cin; ci += 4) {
                const float w0 = wp[(long)(ci + 0) * a.Cout];
                const float w1 = wp[(long)(ci + 1) * a.Cout];
                const float w2 = wp[(long)(ci + 2) * a.Cout];
                const float w3 = wp[(long)(ci + 3) * a.Cout];
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    if (ok[i]) {
                        const int ix = ixb + i * a.stride;
                        const float4 v =
                            *reinterpret_cast<const float4 *>(xrow + (long)ix * a.Cin + ci);
                        acc[i] = fmaf(v.x, w0, acc[i]);
                        acc[i] = fmaf(v.y, w1, acc[i]);
                        acc[i] = fmaf(v.z, w2, acc[i]);
                        acc[i] = fmaf(v.w, w3, acc[i]);
                    }
                }
            }
        }
    }

    const float sc = a.scale ? a.scale[co] : 1.0f;
    const float sh = a.scale ? a.shift[co] : 0.0f;
    const float al = a.alpha ? a.alpha[co] : 0.0f;
    const float ral = a.res_alpha ? a.res_alpha[co] : 0.0f;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int ox = ox0 + i;
        if (ox >= a.Wo) break;
        float v = acc[i];
        if (a.scale) v = fmaf(v, sc, sh);
        if (a.alpha) v = prelu_f(v, al);
        const long opix = ((long)n * a.Ho + oy) * a.Wo + ox;
        if (a.res_mode == RES_ADD) {
            v = v + a.res[opix * a.Cout + co];
        } else if (a.res_mode == RES_POOL) {
            // tf.nn.max_pool_with_argmax 2x2/s2 on the block input, zero-padded in channels at the end
            // (enet_modules.py:927-933).  First maximum in (y,x) order wins (strict '>').
            if (co < a.res_C) {
                const int Hs = 2 * a.Ho, Ws = 2 * a.Wo;
                float best = -FLT_MAX;
                int code = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int iy = 2 * oy + (d >> 1), ix = 2 * ox + (d & 1);
                    const float r = a.res[(((long)n * Hs + iy) * Ws + ix) * a.res_C + co];
                    if (r > best) { best = r; code = d; }
                }
                a.code_out[opix * a.res_C + co] = (uint8_t)code;
                v = v + best;
            }
        } else if (a.res_mode == RES_UNPOOL) {
            // unpool_2d as a gather: every 2x2/s2 pooling index lies inside its own window, so
            // scatter_nd into zeros (extra_ops.py:82-85) == select on the saved window code.
            const int Hs = a.Ho / 2, Ws = a.Wo / 2;
            const long sp = (((long)n * Hs + (oy >> 1)) * Ws + (ox >> 1)) * a.Cout + co;
            const int code = a.code_in[sp];
            const float r = (code == ((oy & 1) * 2 + (ox & 1))) ? a.res[sp] : 0.0f;
            v = v + r;
        }
        if (a.res_alpha) v = prelu_f(v, ral);
        a.y[opix * a.Cout + co] = v;
    }
}

hipError_t launch_conv(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 != 0 || a.Cout <= 0 || a.Cout > 256 || 256 % a.Cout != 0) return hipErrorInvalidValue;
    constexpr int PX = 4;
    const int gx = (a.Wo + PX - 1) / PX;
    const long ngroups = (long)a.N * a.Ho * gx;
    const int gpb = 256 / a.Cout;
    const long grid = (ngroups + gpb - 1) / gpb;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    const double opix = (double)a.N * a.Ho * a.Wo;
    double bytes = 4.0 * ((double)a.N * a.H * a.W * a.Cin + opix * a.Cout + (double)a.KH * a.KW * a.Cin * a.Cout);
    if (a.res_mode == RES_ADD) bytes += 4.0 * opix * a.Cout;
    if (a.res_mode == RES_POOL) bytes += 4.0 * opix * 4 * a.res_C + opix * a.res_C;
    if (a.res_mode == RES_UNPOOL) bytes += (4.0 + 1.0) * opix / 4 * a.Cout;
    ProfScope prof("k_conv", 2.0 * opix * a.Cout * a.KH * a.KW * a.Cin, bytes, s);
    hipLaunchKernelGGL(k_conv<PX>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// tf.nn.conv2d_transpose 3x3 / stride 2 / SAME -> [N,2H,2W,Cout]  (enet_modules.py:1251-1255)
//   out[2i+kh, 2j+kw, o] += in[i,j,c] * W[kh,kw,o,c]; rows/cols 2H / 2W dropped   
// gather form, taps (kh,kw,ci) ascending; wT = [3][3][Cin][Cout]; fused BN + PReLU (nullable).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_convT(const float *__restrict__ x, int N, int H, int W,
                                               int Cin, const float *__restrict__ wT, int Cout,
                                               const float *__restrict__ scale,
                                               const float *__restrict__ shift,
                                               const float *__restrict__ alpha,
                                               float *__restrict__ y)
{
    const int tid = threadIdx.x;
    const int co = tid % Cout;
    const int ppb = 256 / Cout;
    const int Ho = 2 * H, Wo = 2 * W;
    const long npix = (long)N * Ho * Wo;
    const long p = (long)blockIdx.x * ppb + tid / Cout;
    if (p >= npix) return;
    const int ox = (int)(p % Wo);
    const int oy = (int)((p / Wo) % Ho);
    const int n = (int)(p / ((long)Wo * Ho));
    float acc = 0.0f;
    for (int kh = 0; kh < 3; ++kh) {
        const int ty = oy - kh;
        if (ty < 0 || (ty & 1)) continue;
        const int iy = ty >> 1;
        if (iy >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int tx = ox - kw;
            if (tx < 0 || (tx & 1)) continue;
            const int ix = tx >> 1;
            if (ix >= W) continue;
            const float *xp = x + (((long)n * H + iy) * W + ix) * Cin;
            const float *wp = wT + (long)((kh * 3 + kw) * Cin) * Cout + co;
            for (int ci = 0; ci < Cin; ci += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(xp + ci);
                acc = fmaf(v.x, wp[(long)(ci + 0) * Cout], acc);
                acc = fmaf(v.y, wp[(long)(ci + 1) * Cout], acc);
                acc = fmaf(v.z, wp[(long)(ci + 2) * Cout], acc);
                acc = fmaf(v.w, wp[(long)(ci + 3) * Cout], acc);
            }
        }
    }
    float v = acc;
    if (scale) v = fmaf(v, scale[co], shift[co]);
    if (alpha) v = prelu_f(v, alpha[co]);
    y[p * Cout + co] = v;
}

hipError_t launch_convT(const float *x, int N, int H, int W, int Cin, const float *wT, int Cout,
                        const float *scale, const float *shift, const float *alpha, float *y,
                        hipStream_t s)
{
    if (Cin % 4 != 0 || Cout <= 0 || Cout > 256 || 256 % Cout != 0) return hipErrorInvalidValue;
    const long npix = (long)N * 4 * H * W;
    const int ppb = 256 / Cout;
    const long grid = (npix + ppb - 1) / ppb;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    ProfScope prof("k_convT", 2.0 * (double)N * H * W * 9 * Cin * Cout,
                   4.0 * ((double)N * H * W * Cin + (double)npix * Cout + 9.0 * Cin * Cout), s);
    hipLaunchKernelGGL(k_convT, dim3((unsigned)grid), dim3(256), 0, s, x, N, H, W, Cin, wT, Cout,
                       scale, shift, alpha, y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Final (conv2d_transpose 3x3 s2, 16 -> K, no bias/BN/activation; enet_modules.py:1359-1381) fused
// with the score.  One thread per INPUT pixel (i,j) = one 2x2 output quad, so the kernel taps of
// every output parity are wave-uniform (scalar loads) and there is no divergence:
//   out(2i  ,2j  ) = a*W00 + c*W02 + b*W20 + d*W22     a=in(i,j) b=in(i-1,j) c=in(i,j-1) d=in(i-1,j-1)
//   out(2i  ,2j+1) = a*W01 + b*W21
//   out(2i+1,2j  ) = a*W10 + c*W12
//   out(2i+1,2j+1) = a*W11                      (taps listed in (kh,kw) ascending = oracle order)
// wF = [3][3][16][K2], K2 = K rounded up to even (zero padded at commit): two classes per
// v_pk_fma_f32 -- the kernel tap pair sits in an aligned SGPR pair, the activation is broadcast to both
// halves (op_sel), each half is an ordinary IEEE fma, so the chain per class is unchanged.
// grid = (ceil(H/16) * ceil(W/16), N): a workgroup owns a 16 x 16 tile of input pixels of one image, staged through LDS.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One kernel tap (16 input channels x K classes) applied to the class-pair accumulators.  The taps are
// wave-uniform: they are read with scalar loads and enter v_pk_fma_f32 as SGPR pairs.  The 9 x 16 x K2 of
// them do not fit the SGPR file, so they stream through two buffers of G input channels each: the scalar
// loads of group g+1 (or of the first group of the NEXT tap, wnext) are issued before the FMAs of group
// g; sched_barrier pins that order.  (Left alone the compiler hoists whole taps and spills ~1500 SGPRs
// through v_writelane / v_readlane, doubling the VALU work of this VALU-bound kernel.)
// Scalar loads return out of order, so the only wait the hardware offers is lgkmcnt(0) = "everything": the
// explicit wait at the top of every step completes the group about to be used BEFORE the next prefetch is
// issued (a compiler-placed wait would sit after it and drain the prefetch as well, i.e. expose the full
// scalar-load latency at every step).  G = input channels per group: the FMAs of one group (10 G
// instructions for K = 19) are all the cover a prefetch gets, so G = 2 where the SGPR file allows it (the
// score-only kernel; with the optional outputs' pointers live, 80 buffer SGPRs spill).
// On entry w0 holds group 0 of this tap; on exit it holds group 0 of wnext (if not NULL).
template <int K, int G>
struct FsTap {
    static constexpr int KP = (K + 1) / 2;           // class pairs
    static constexpr int NG = 16 / G, WN = G * 2 * KP, TS = 16 * 2 * KP;
    static constexpr int kWaitScalar = 0xC07F;        // s_waitcnt lgkmcnt(0), vmcnt / expcnt untouched
    static __device__ __forceinline__ void load(const float *__restrict__ p, float (&w)[WN])
    {
#pragma unroll
        for (int q = 0; q < WN; ++q) w[q] = p[q];
    }
    static __device__ __forceinline__ void fma(f32x2 (&acc)[KP], const float (&v)[16], int c0, const float (&w)[WN])
    {
#pragma unroll
        for (int c = 0; c < G; ++c) {
            const f32x2 a2 = {v[c0 + c], v[c0 + c]};
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const f32x2 w2 = {w[c * 2 * KP + 2 * p], w[c * 2 * KP + 2 * p + 1]};
                acc[p] = __builtin_elementwise_fma(a2, w2, acc[p]);
            }
        }
    }
    static __device__ __forceinline__ void apply(f32x2 (&acc)[KP], const float (&v)[16],
                                                 const float *__restrict__ wtap, const float *__restrict__ wnext,
                                                 float (&w0)[WN], float (&w1)[WN])
    {
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            __builtin_amdgcn_s_waitcnt(kWaitScalar);  // w0 (requested one step ago) has arrived
            __builtin_amdgcn_sched_barrier(0);
            load(wtap + (g + 1) * WN, w1);
            __builtin_amdgcn_sched_barrier(0);
            fma(acc, v, g * G, w0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(kWaitScalar);  // w1 has arrived
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) load(wtap + (g + 2) * WN, w0);
            else if (wnext) load(wnext, w0);
            __builtin_amdgcn_sched_barrier(0);
            fma(acc, v, (g + 1) * G, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// Workgroup = 16 x 16 input pixels (= 32 x 32 output pixels) of one image; FS_PS = LDS floats per pixel (16 + 4 pad:
// the ds_read_b128 of 16 neighbouring pixels then fall on 16 different bank quads).
constexpr int FS_T = 16, FS_TP = FS_T + 1, FS_PS = 20;

// Bottleneck5_1 (width-4 regular bottleneck on 16 channels, enet_modules.py:526-599) evaluated inside the Final + score
// kernel (F51 instantiations): the kernel's 17 x 17 input window is then COMPUTED from Bottleneck5_0's output instead of
// loaded -- 5_1's 268 MB output (batch 8 x 1024 x 2048) is neither written nor read, and its launch (HBM-bound, its
// width-4 GEMMs zero-padded to 16 MFMA columns) disappears.  272 MACs per pixel on wave-uniform (scalar-load) kernels:
//   phase P  projection 16 -> 4 + BN + PReLU on the 19 x 19 window (exact zeros outside the image: SAME padding applies
//            to the PROJECTED tensor) -> LDS
//   phase C  3x3 conv 4 -> 4 + BN + PReLU, expansion 4 -> 16 + BN, + x, PReLU on the 17 x 17 window -> the LDS tile the
//            transposed convolution reads (zeros outside the image, as the staged load writes them)
// every chain in the order of k_bottleneck16 / the oracle ((kh, kw, ci) ascending fmaf, y = fmaf(acc, s, t)).
struct Bnk4Args {
    const float *x5;                 // Bottleneck5_0 output [N,H,W,16]
    const float *wp, *ps, *pt, *pa;  // proj kernel [16][4], folded BN, alpha
    const float *wc, *cs, *ct, *ca;  // conv kernel [3][3][4][4]
    const float *we, *es, *et, *ra;  // exp kernel [4][16], folded BN [16], residual alpha [16]
};
constexpr int F51_PW = FS_TP + 2;    // projected window: 19 x 19

// OUT = false: score only (the ranking pass): logits / label / mask / conf are not touched, which frees the
// SGPRs their pointers would pin and lets the kernel taps stream two input channels at a time.
template <int K, bool OUT, bool F51 = false>
__global__ __launch_bounds__(256) void k_final_score(const float *__restrict__ x, int N, int H,
                                                     int W, const float *__restrict__ wF,
                                                     float *__restrict__ logits, int measure,
                                                     float threshold, double *__restrict__ partial,
                                                     uint8_t *__restrict__ label,
                                                     uint8_t *__restrict__ mask,
                                                     float *__restrict__ conf, Bnk4Args b5)
{
    __shared__ double red[4];
    __shared__ __attribute__((aligned(16))) float tile[FS_TP * FS_TP * FS_PS];
    __shared__ __attribute__((aligned(16))) float p1[F51 ? F51_PW * F51_PW * 4 : 4];
    const int n = blockIdx.y;
    const long HW = (long)H * W;
    const int tiles_x = (W + FS_T - 1) / FS_T;
    const int i0 = (int)(blockIdx.x / tiles_x) * FS_T, j0 = (int)(blockIdx.x % tiles_x) * FS_T;
    if (F51) {
        const float *x5 = b5.x5 + (long)n * HW * 16;
        // ---- phase P: both passes' loads are requested first (361 pixels over 256 threads)
        float4 xin[2][4];
        bool okp[2];
#pragma unroll
        for (int ps_ = 0; ps_ < 2; ++ps_) {
            const int e = min((int)threadIdx.x + 256 * ps_, F51_PW * F51_PW - 1);
            const int gi = i0 - 2 + e / F51_PW, gj = j0 - 2 + e % F51_PW;
            okp[ps_] = gi >= 0 && gi < H && gj >= 0 && gj < W;
            const float4 *xp = reinterpret_cast<const float4 *>(x5 + ((long)min(max(gi, 0), H - 1) * W + min(max(gj, 0), W - 1)) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) xin[ps_][q] = xp[q];
        }
#pragma unroll
        for (int ps_ = 0; ps_ < 2; ++ps_) {
            const int e = (int)threadIdx.x + 256 * ps_;
            if (e < F51_PW * F51_PW) {  // wave-uniform except in the last active wave
                const float xv[16] = {xin[ps_][0].x, xin[ps_][0].y, xin[ps_][0].z, xin[ps_][0].w, xin[ps_][1].x, xin[ps_][1].y,
                                      xin[ps_][1].z, xin[ps_][1].w, xin[ps_][2].x, xin[ps_][2].y, xin[ps_][2].z, xin[ps_][2].w,
                                      xin[ps_][3].x, xin[ps_][3].y, xin[ps_][3].z, xin[ps_][3].w};
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ci = 0; ci < 16; ++ci)
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f] = fmaf(xv[ci], b5.wp[ci * 4 + f], acc[f]);
                float4 o;
                o.x = okp[ps_] ? prelu_f(fmaf(acc[0], b5.ps[0], b5.pt[0]), b5.pa[0]) : 0.0f;
                o.y = okp[ps_] ? prelu_f(fmaf(acc[1], b5.ps[1], b5.pt[1]), b5.pa[1]) : 0.0f;
                o.z = okp[ps_] ? prelu_f(fmaf(acc[2], b5.ps[2], b5.pt[2]), b5.pa[2]) : 0.0f;
                o.w = okp[ps_] ? prelu_f(fmaf(acc[3], b5.ps[3], b5.pt[3]), b5.pa[3]) : 0.0f;
                reinterpret_cast<float4 *>(p1)[e] = o;
            }
        }
        __syncthreads();
        // ---- phase C: 289 pixels over 256 threads; the residual rows (L1 / L2 hits: phase P has just read them) first
        float4 xr[2][4];
        bool okc[2];
#pragma unroll
        for (int ps_ = 0; ps_ < 2; ++ps_) {
            const int e = min((int)threadIdx.x + 256 * ps_, FS_TP * FS_TP - 1);
            const int gi = i0 - 1 + e / FS_TP, gj = j0 - 1 + e % FS_TP;
            okc[ps_] = gi >= 0 && gi < H && gj >= 0 && gj < W;
            const float4 *xp = reinterpret_cast<const float4 *>(x5 + ((long)min(max(gi, 0), H - 1) * W + min(max(gj, 0), W - 1)) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[ps_][q] = xp[q];
        }
#pragma unroll
        for (int ps_ = 0; ps_ < 2; ++ps_) {
            const int e = (int)threadIdx.x + 256 * ps_;
            if (e < FS_TP * FS_TP) {
                const int pi = e / FS_TP, pj = e % FS_TP;
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float4 pv = reinterpret_cast<const float4 *>(p1)[(pi + kh) * F51_PW + (pj + kw)];
                        const float pc[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                            for (int f = 0; f < 4; ++f) acc[f] = fmaf(pc[ci], b5.wc[((kh * 3 + kw) * 4 + ci) * 4 + f], acc[f]);
                    }
                float qv[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) qv[f] = prelu_f(fmaf(acc[f], b5.cs[f], b5.ct[f]), b5.ca[f]);
                const float xres[16] = {xr[ps_][0].x, xr[ps_][0].y, xr[ps_][0].z, xr[ps_][0].w, xr[ps_][1].x, xr[ps_][1].y,
                                        xr[ps_][1].z, xr[ps_][1].w, xr[ps_][2].x, xr[ps_][2].y, xr[ps_][2].z, xr[ps_][2].w,
                                        xr[ps_][3].x, xr[ps_][3].y, xr[ps_][3].z, xr[ps_][3].w};
                float out[16];
#pragma unroll
                for (int co = 0; co < 16; ++co) {
                    float ev = 0.0f;
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci) ev = fmaf(qv[ci], b5.we[ci * 16 + co], ev);
                    const float v = prelu_f(fmaf(ev, b5.es[co], b5.et[co]) + xres[co], b5.ra[co]);
                    out[co] = okc[ps_] ? v : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(tile + e * FS_PS + 4 * q) = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
            }
        }
    } else {
    // ---- stage the (16+1) x (16+1) pixel window (one halo row above, one halo column to the left) through LDS:
        // every input pixel is fetched once per workgroup with coalesced float4 loads; outside the image = zeros
        // (the transposed conv has no contribution from there).  All loads are issued before the first LDS write.
        {
            constexpr int NQ = FS_TP * FS_TP * 4, IT = (NQ + 255) / 256;  // float4 quads of the window
            float4 st[IT];
    #pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int e = min((int)threadIdx.x + 256 * it, NQ - 1);
                const int q = e & 3, pj = (e >> 2) % FS_TP, pi = (e >> 2) / FS_TP;
                const int gi = i0 - 1 + pi, gj = j0 - 1 + pj;
                const bool ok = gi >= 0 && gi < H && gj >= 0 && gj < W;
                const long gp = (long)min(max(gi, 0), H - 1) * W + min(max(gj, 0), W - 1);
                const float4 t = reinterpret_cast<const float4 *>(x + ((long)n * HW + gp) * 16)[q];
                st[it] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    #pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int e = (int)threadIdx.x + 256 * it;
                if (e < NQ) *reinterpret_cast<float4 *>(tile + (e >> 2) * FS_PS + 4 * (e & 3)) = st[it];
            }
        }
}
    __syncthreads();
    const int ti = threadIdx.x / FS_T, tj = threadIdx.x % FS_T;
    const int i = i0 + ti, j = j0 + tj;
    const bool valid = i < H && j < W;
    double local = 0.0;
    if (valid) {
        float va[16], vb[16], vc[16], vd[16];
        const float *la = tile + ((ti + 1) * FS_TP + tj + 1) * FS_PS;  // own pixel; b = above, c = left, d = above-left
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 t = reinterpret_cast<const float4 *>(la)[q];
            va[4 * q] = t.x; va[4 * q + 1] = t.y; va[4 * q + 2] = t.z; va[4 * q + 3] = t.w;
            t = reinterpret_cast<const float4 *>(la - FS_TP * FS_PS)[q];
            vb[4 * q] = t.x; vb[4 * q + 1] = t.y; vb[4 * q + 2] = t.z; vb[4 * q + 3] = t.w;
            t = reinterpret_cast<const float4 *>(la - FS_PS)[q];
            vc[4 * q] = t.x; vc[4 * q + 1] = t.y; vc[4 * q + 2] = t.z; vc[4 * q + 3] = t.w;
            t = reinterpret_cast<const float4 *>(la - FS_TP * FS_PS - FS_PS)[q];
            vd[4 * q] = t.x; vd[4 * q + 1] = t.y; vd[4 * q + 2] = t.z; vd[4 * q + 3] = t.w;
        }
        const float inv_logK = 1.0f / __logf((float)K);
        const int Wo = 2 * W;
        typedef FsTap<K, (!OUT && 2 * ((K + 1) / 2) <= 20) ? 2 : 1> FT;
        constexpr int KP = FT::KP, TS = FT::TS;
        float w0[FT::WN], w1[FT::WN];
        f32x2 acc2[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) acc2[k] = (f32x2){0.0f, 0.0f};
        auto finish_quad = [&](int quad) {
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = (k & 1) ? acc2[k >> 1].y : acc2[k >> 1].x;
#pragma unroll
            for (int k = 0; k < KP; ++k) acc2[k] = (f32x2){0.0f, 0.0f};
            const int oy = 2 * i + (quad >> 1), ox = 2 * j + (quad & 1);
            const long op = ((long)n * 2 * H + oy) * Wo + ox;
            if (OUT && logits) {
                float *lp = logits + op * K;
#pragma unroll
                for (int k = 0; k < K; ++k) lp[k] = acc[k];
            }
            int lab;
            const float cf = pixel_score<K>(acc, measure, inv_logK, lab);
            local += (double)cf;
            if (OUT && label) label[op] = (uint8_t)lab;
            if (OUT && mask) mask[op] = cf < threshold ? (uint8_t)0 : (uint8_t)1;
            if (OUT && conf) conf[op] = cf;
        };
        auto tap = [&](int kh, int kw) { return wF + (kh * 3 + kw) * TS; };
        FT::load(tap(0, 0), w0);
        // (even, even): taps (0,0) a, (0,2) c, (2,0) b, (2,2) d
        FT::apply(acc2, va, tap(0, 0), tap(0, 2), w0, w1);
        FT::apply(acc2, vc, tap(0, 2), tap(2, 0), w0, w1);
        FT::apply(acc2, vb, tap(2, 0), tap(2, 2), w0, w1);
        FT::apply(acc2, vd, tap(2, 2), tap(0, 1), w0, w1);
        finish_quad(0);
        // (even, odd): (0,1) a, (2,1) b
        FT::apply(acc2, va, tap(0, 1), tap(2, 1), w0, w1);
        FT::apply(acc2, vb, tap(2, 1), tap(1, 0), w0, w1);
        finish_quad(1);
        // (odd, even): (1,0) a, (1,2) c
        FT::apply(acc2, va, tap(1, 0), tap(1, 2), w0, w1);
        FT::apply(acc2, vc, tap(1, 2), tap(1, 1), w0, w1);
        finish_quad(2);
        // (odd, odd): (1,1) a
        FT::apply(acc2, va, tap(1, 1), nullptr, w0, w1);
        finish_quad(3);
    }
    const double r = block_sum_256(local, red);
    if (threadIdx.x == 0) partial[(long)n * gridDim.x + blockIdx.x] = r;
}

int final_score_blocks(int H, int W) { return cdiv(H, FS_T) * cdiv(W, FS_T); }

// Bottleneck5_1 + Final + score (score-only form): x5 = Bottleneck5_0's output
hipError_t launch_bnk4_final_score(const float *x5, int N, int H, int W, const float *wp, const float *ps, const float *pt,
                                   const float *pa, const float *wc, const float *cs, const float *ct, const float *ca,
                                   const float *we, const float *es, const float *et, const float *ra, const float *wF, int K,
                                   int measure, double *partial, hipStream_t s)
{
    dim3 grid(final_score_blocks(H, W), N), block(256);
    const double pix = (double)N * H * W;
    ProfScope prof("k_final_score<fused 5_1>", 2.0 * pix * 9 * 16 * K + 2.0 * pix * (16.0 * 4 + 9.0 * 4 * 4 + 4.0 * 16),
                   4.0 * pix * 16, s);
    Bnk4Args b = {x5, wp, ps, pt, pa, wc, cs, ct, ca, we, es, et, ra};
#define SSAL_FS51(KK)                                                                                               \
    case KK:                                                                                                        \
        hipLaunchKernelGGL((k_final_score<KK, false, true>), grid, block, 0, s, nullptr, N, H, W, wF, nullptr, measure, \
                           0.0f, partial, nullptr, nullptr, nullptr, b);                                            \
        break;
    switch (K) {
        SSAL_FS51(2) SSAL_FS51(3) SSAL_FS51(4) SSAL_FS51(5) SSAL_FS51(6) SSAL_FS51(7) SSAL_FS51(8) SSAL_FS51(9)
        SSAL_FS51(10) SSAL_FS51(11) SSAL_FS51(12) SSAL_FS51(13) SSAL_FS51(14) SSAL_FS51(15) SSAL_FS51(16)
        SSAL_FS51(17) SSAL_FS51(18) SSAL_FS51(19) SSAL_FS51(20) SSAL_FS51(21) SSAL_FS51(22) SSAL_FS51(23)
        SSAL_FS51(24) SSAL_FS51(25) SSAL_FS51(26) SSAL_FS51(27) SSAL_FS51(28) SSAL_FS51(29) SSAL_FS51(30)
        SSAL_FS51(31) SSAL_FS51(32)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_FS51
    return hipGetLastError();
}

hipError_t launch_final_score(const float *x, int N, int H, int W, const float *wF, int K,
                              float *logits, int measure, float threshold, double *partial,
                              uint8_t *label, uint8_t *mask, float *conf, hipStream_t s)
{
    dim3 grid(final_score_blocks(H, W), N), block(256);
    const Bnk4Args b = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ProfScope prof("k_final_score", 2.0 * (double)N * H * W * 9 * 16 * K,
                   4.0 * ((double)N * H * W * 16 + (logits ? (double)N * 4 * H * W * K : 0.0)) +
                       (double)N * 4 * H * W * ((label ? 1 : 0) + (mask ? 1 : 0) + (conf ? 4 : 0)), s);
    const bool out = logits || label || mask || conf;
#define SSAL_FS(KK)                                                                                        \
    case KK:                                                                                               \
        if (out)                                                                                           \
            hipLaunchKernelGGL((k_final_score<KK, true>), grid, block, 0, s, x, N, H, W, wF, logits, measure, \
                               threshold, partial, label, mask, conf, b);                                  \
        else                                                                                               \
            hipLaunchKernelGGL((k_final_score<KK, false>), grid, block, 0, s, x, N, H, W, wF, logits, measure, \
                               threshold, partial, label, mask, conf, b);                                  \
        break;
    switch (K) {
        SSAL_FS(2) SSAL_FS(3) SSAL_FS(4) SSAL_FS(5) SSAL_FS(6) SSAL_FS(7) SSAL_FS(8) SSAL_FS(9)
        SSAL_FS(10) SSAL_FS(11) SSAL_FS(12) SSAL_FS(13) SSAL_FS(14) SSAL_FS(15) SSAL_FS(16)
        SSAL_FS(17) SSAL_FS(18) SSAL_FS(19) SSAL_FS(20) SSAL_FS(21) SSAL_FS(22) SSAL_FS(23)
        SSAL_FS(24) SSAL_FS(25) SSAL_FS(26) SSAL_FS(27) SSAL_FS(28) SSAL_FS(29) SSAL_FS(30)
        SSAL_FS(31) SSAL_FS(32)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_FS
    return hipGetLastError();
}

// scores[n] = (sum of partial[n,:]) / pixels; one block per image; fixed order => reproducible.
__global__ __launch_bounds__(256) void k_reduce_mean(const double *__restrict__ partial, int blocks,
                                                     double pixels, double *__restrict__ scores)
{
    __shared__ double red[4];
    const int n = blockIdx.x;
    double v = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) v += partial[(long)n * blocks + b];
    const double r = block_sum_256(v, red);
    if (threadIdx.x == 0) scores[n] = r / pixels;
}

hipError_t launch_reduce_mean(const double *partial, int N, int blocks, double pixels,
                              double *scores, hipStream_t s)
{
    hipLaunchKernelGGL(k_reduce_mean, dim3(N), dim3(256), 0, s, partial, blocks, pixels, scores);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Stand-alone score on materialised logits [N,H,W,K] (HBM-bound: K*4 B read per pixel, ~0 written).
// A block stages 256 pixels x K floats through LDS with coalesced float4 loads; each thread then
// reads its own pixel's K logits at stride K dwords (bank-conflict-free for odd K, 2-way for even).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_score_logits(const float *__restrict__ logits, int N,
                                                      long HW, int measure, float threshold,
                                                      double *__restrict__ partial,
                                                      uint8_t *__restrict__ label,
                                                      uint8_t *__restrict__ mask,
                                                      float *__restrict__ conf)
{
    constexpr int KP = (K % 2 == 0) ? K + 1 : K;  // odd LDS stride
    __shared__ float tile[256 * KP];
    __shared__ double red[4];
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * 256;
    const int npx = (int)((HW - p0) < 256 ? (HW - p0) : 256);
    const float *src = logits + ((long)n * HW + p0) * K;
    const int nflt = npx * K;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;  // block-uniform
    for (int f = threadIdx.x * 4; f < nflt; f += 1024) {
        if (vec_ok && f + 3 < nflt) {
            const float4 t = *reinterpret_cast<const float4 *>(src + f);
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = f + q;
                tile[(e / K) * KP + (e % K)] = tv[q];
            }
        } else {
            for (int e = f; e < nflt && e < f + 4; ++e) tile[(e / K) * KP + (e % K)] = src[e];
        }
    }
    __syncthreads();
    double local = 0.0;
    if ((int)threadIdx.x < npx) {
        float l[K];
#pragma unroll
        for (int k = 0; k < K; ++k) l[k] = tile[threadIdx.x * KP + k];
        int lab;
        const float cf = pixel_score<K>(l, measure, 1.0f / __logf((float)K), lab);
        local = (double)cf;
        const long op = (long)n * HW + p0 + threadIdx.x;
        if (label) label[op] = (uint8_t)lab;
        if (mask) mask[op] = cf < threshold ? (uint8_t)0 : (uint8_t)1;
        if (conf) conf[op] = cf;
    }
    const double r = block_sum_256(local, red);
    if (threadIdx.x == 0) partial[(long)n * gridDim.x + blockIdx.x] = r;
}

int score_blocks(int H, int W) { return cdiv((long)H * W, 256); }

hipError_t launch_score_logits(const float *logits, int N, int H, int W, int K, int measure,
                               float threshold, double *partial, uint8_t *label, uint8_t *mask,
                               float *conf, hipStream_t s)
{
    dim3 grid(score_blocks(H, W), N), block(256);
    const long HW = (long)H * W;
    ProfScope prof("k_score_logits", (double)N * HW * K * 6.0,
                   (double)N * HW * (4.0 * K + (label ? 1 : 0) + (mask ? 1 : 0) + (conf ? 4 : 0)), s);
#define SSAL_SL(KK)                                                                              \
    case KK:                                                                                     \
        hipLaunchKernelGGL(k_score_logits<KK>, grid, block, 0, s, logits, N, HW, measure,        \
                           threshold, partial, label, mask, conf);                               \
        break;
    switch (K) {
        SSAL_SL(2) SSAL_SL(3) SSAL_SL(4) SSAL_SL(5) SSAL_SL(6) SSAL_SL(7) SSAL_SL(8) SSAL_SL(9)
        SSAL_SL(10) SSAL_SL(11) SSAL_SL(12) SSAL_SL(13) SSAL_SL(14) SSAL_SL(15) SSAL_SL(16)
        SSAL_SL(17) SSAL_SL(18) SSAL_SL(19) SSAL_SL(20) SSAL_SL(21) SSAL_SL(22) SSAL_SL(23)
        SSAL_SL(24) SSAL_SL(25) SSAL_SL(26) SSAL_SL(27) SSAL_SL(28) SSAL_SL(29) SSAL_SL(30)
        SSAL_SL(31) SSAL_SL(32)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_SL
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// masked_softmax_cross_entropy forward (tensortools/losses.py:3-74):
//   y_k = one_hot(label, K, on = 1 - ls, off = ls / (K - 1));   ce = sum_k y_k * (log S - (x_k - m))
//   ce *= mask;  if weight > 1: ce *= 1 / log(weight + (1.718281828459045 - weight) * sum_k p_k y_k)
//   loss = sum_pos (double)(sum_n ce[n,pos]) / (double)(float)(sum mask)      (fp32 over the batch axis,
//   float64 over the spatial axes, :62-73).  Block partials + fixed-order final sum (reproducible).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_masked_xent(const float *__restrict__ logits,
                                                     const uint8_t *__restrict__ labels,
                                                     const float *__restrict__ mask, int N, long HW,
                                                     float weight, float on_value, float off_value,
                                                     double *__restrict__ partial)
{
    __shared__ double red[4];
    const long pos = (long)blockIdx.x * 256 + threadIdx.x;
    double loss = 0.0, msum = 0.0;
    if (pos < HW) {
        float bsum = 0.0f;  // tf.reduce_sum(loss, axis=0) in fp32
        for (int n = 0; n < N; ++n) {
            const float *l = logits + ((long)n * HW + pos) * K;
            const int lab = labels[(long)n * HW + pos];
            const float mk = mask[(long)n * HW + pos];
            float x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) x[k] = l[k];
            float m = x[0];
#pragma unroll
            for (int k = 1; k < K; ++k) m = fmaxf(m, x[k]);
            float S = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) S += expf(x[k] - m);
            const float logS = logf(S);
            float ce = 0.0f, pc = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float yk = (k == lab) ? on_value : off_value;
                const float d = x[k] - m;
                ce += yk * (logS - d);
                pc += yk * (expf(d) / S);
            }
            ce *= mk;
            if (weight > 1.0f) ce *= 1.0f / logf(weight + (1.718281828459045f - weight) * pc);
            bsum += ce;
            msum += (double)mk;
        }
        loss = (double)bsum;
    }
    const double r0 = block_sum_256(loss, red);
    __syncthreads();
    const double r1 = block_sum_256(msum, red);
    if (threadIdx.x == 0) {
        partial[2 * (long)blockIdx.x] = r0;
        partial[2 * (long)blockIdx.x + 1] = r1;
    }
}

__global__ __launch_bounds__(256) void k_xent_finish(const double *__restrict__ partial, int blocks,
                                                     double *__restrict__ out)
{
    __shared__ double red[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) { a += partial[2 * (long)i]; b += partial[2 * (long)i + 1]; }
    const double ra = block_sum_256(a, red);
    __syncthreads();
    const double rb = block_sum_256(b, red);
    if (threadIdx.x == 0) out[0] = ra / (double)(float)rb;  // tf.cast(tf.reduce_sum(_mask) [fp32], float64)
}

int xent_blocks(int H, int W) { return cdiv((long)H * W, 256); }

hipError_t launch_masked_xent(const float *logits, const uint8_t *labels, const float *mask, int N,
                              int H, int W, int K, float weight, float label_smoothing,
                              double *partial, double *out, hipStream_t s)
{
    const long HW = (long)H * W;
    const int blocks = xent_blocks(H, W);
    const float on_value = 1.0f - label_smoothing, off_value = label_smoothing / ((float)K - 1.0f);
#define SSAL_XE(KK)                                                                               \
    case KK:                                                                                      \
        hipLaunchKernelGGL(k_masked_xent<KK>, dim3(blocks), dim3(256), 0, s, logits, labels, mask, \
                           N, HW, weight, on_value, off_value, partial);                          \
        break;
    switch (K) {
        SSAL_XE(2) SSAL_XE(3) SSAL_XE(4) SSAL_XE(5) SSAL_XE(6) SSAL_XE(7) SSAL_XE(8) SSAL_XE(9)
        SSAL_XE(10) SSAL_XE(11) SSAL_XE(12) SSAL_XE(13) SSAL_XE(14) SSAL_XE(15) SSAL_XE(16)
        SSAL_XE(17) SSAL_XE(18) SSAL_XE(19) SSAL_XE(20) SSAL_XE(21) SSAL_XE(22) SSAL_XE(23)
        SSAL_XE(24) SSAL_XE(25) SSAL_XE(26) SSAL_XE(27) SSAL_XE(28) SSAL_XE(29) SSAL_XE(30)
        SSAL_XE(31) SSAL_XE(32)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_XE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_xent_finish, dim3(1), dim3(256), 0, s, partial, blocks, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// tf.nn.max_pool_with_argmax 2x2/s2 with the reference's int64 index and the scatter
// form of xops.unpool_2d (extra_ops.py:28-86) for arbitrary index tensors.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool_argmax(const float *__restrict__ x, int N, int H,
                                                        int W, int C, float *__restrict__ y,
                                                        int64_t *__restrict__ argmax,
                                                        int include_batch)
{
    const int Ho = H / 2, Wo = W / 2;
    const long total = (long)N * Ho * Wo * C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C);
        const long pix = o / C;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((long)Wo * Ho));
        float best = -FLT_MAX;
        long bi = -1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int iy = 2 * oy + (d >> 1), ix = 2 * ox + (d & 1);
            const float v = x[(((long)n * H + iy) * W + ix) * C + c];
            if (v > best) { best = v; bi = ((long)iy * W + ix) * C + c; }
        }
        y[o] = best;
        if (argmax) argmax[o] = bi + (include_batch ? (long)n * H * W * C : 0L);
    }
}

hipError_t launch_maxpool_argmax(const float *x, int N, int H, int W, int C, float *y,
                                 int64_t *argmax, int include_batch, hipStream_t s)
{
    const long total = (long)N * (H / 2) * (W / 2) * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_maxpool_argmax, dim3(grid), dim3(256), 0, s, x, N, H, W, C, y, argmax,
                       include_batch);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_unpool_scatter(const float *__restrict__ x,
                                                        const int64_t *__restrict__ idx, long per_img_in,
                                                        long total_in, long per_img_out, long total_out,
                                                        int idx_has_batch, float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_in; i += (long)gridDim.x * 256) {
        const long n = i / per_img_in;
        const long k = idx[i] + (idx_has_batch ? 0L : n * per_img_out);
        if (k >= 0 && k < total_out) y[k] = x[i];  // out-of-range indices are dropped, never written
    }
}

hipError_t launch_unpool_scatter(const float *x, const int64_t *idx, int N, int H, int W, int C,
                                 int idx_has_batch, float *y, hipStream_t s)
{
    const long per_in = (long)H * W * C, per_out = 4 * per_in;
    hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * per_out * N, s);
    if (e != hipSuccess) return e;
    int grid = cdiv(per_in * N, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_unpool_scatter, dim3(grid), dim3(256), 0, s, x, idx, per_in, per_in * N,
                       per_out, per_out * N, idx_has_batch, y);
    return hipGetLastError();
}

// window code (dy*2+dx) at pooled position -> reference int64 index into the un-pooled [2Ho,2Wo,C] image
__global__ __launch_bounds__(256) void k_codes_to_argmax(const uint8_t *__restrict__ code, int N,
                                                         int Ho, int Wo, int C,
                                                         int64_t *__restrict__ argmax)
{
    const long total = (long)N * Ho * Wo * C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C);
        const long pix = o / C;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const int d = code[o];
        argmax[o] = ((long)(2 * oy + (d >> 1)) * (2 * Wo) + (2 * ox + (d & 1))) * C + c;
    }
}

hipError_t launch_codes_to_argmax(const uint8_t *code, int N, int Ho, int Wo, int C,
                                  int64_t *argmax, hipStream_t s)
{
    const long total = (long)N * Ho * Wo * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_codes_to_argmax, dim3(grid), dim3(256), 0, s, code, N, Ho, Wo, C, argmax);
    return hipGetLastError();
}

// reference int64 index -> window code; *bad is set if an index does not lie inside its own 2x2 window
// (then the caller must fall back to the scatter form of unpool_2d)
__global__ __launch_bounds__(256) void k_argmax_to_codes(const int64_t *__restrict__ argmax, int N,
                                                         int Ho, int Wo, int C,
                                                         uint8_t *__restrict__ code,
                                                         int *__restrict__ bad)
{
    const long total = (long)N * Ho * Wo * C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C);
        const long pix = o / C;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long idx = argmax[o];
        int found = -1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const long cand = ((long)(2 * oy + (d >> 1)) * (2 * Wo) + (2 * ox + (d & 1))) * C + c;
            if (idx == cand) found = d;
        }
        if (found < 0) { atomicOr(bad, 1); found = 0; }
        code[o] = (uint8_t)found;
    }
}

hipError_t launch_argmax_to_codes(const int64_t *argmax, int N, int Ho, int Wo, int C, uint8_t *code,
                                  int *bad, hipStream_t s)
{
    const long total = (long)N * Ho * Wo * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_argmax_to_codes, dim3(grid), dim3(256), 0, s, argmax, N, Ho, Wo, C, code, bad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Elementwise operators of models/util/extra_ops.py exposed through the C ABI.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prelu(const float *__restrict__ x, long total, int C,
                                               const float *__restrict__ alpha,
                                               float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
        y[i] = prelu_f(x[i], alpha[i % C]);
}

hipError_t launch_prelu(const float *x, int64_t pixels, int C, const float *alpha, float *y,
                        hipStream_t s)
{
    const long total = pixels * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_prelu, dim3(grid), dim3(256), 0, s, x, total, C, alpha, y);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_affine(const float *__restrict__ x, long total, int C,
                                                const float *__restrict__ scale,
                                                const float *__restrict__ shift,
                                                float *__restrict__ y)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        y[i] = fmaf(x[i], scale[c], shift[c]);
    }
}

hipError_t launch_affine(const float *x, int64_t pixels, int C, const float *scale,
                         const float *shift, float *y, hipStream_t s)
{
    const long total = pixels * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_affine, dim3(grid), dim3(256), 0, s, x, total, C, scale, shift, y);
    return hipGetLastError();
}

// tf.nn.fused_batch_norm(is_training=False), eps = 1e-3 (extra_ops.py:181-184) folded to (scale, shift)
__global__ void k_bn_fold(const float *__restrict__ mean, const float *__restrict__ var,
                          const float *__restrict__ gamma, const float *__restrict__ beta, int C,
                          float *__restrict__ scale, float *__restrict__ shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sgm = gamma[c] / sqrtf(var[c] + 1e-3f);
        scale[c] = sgm;
        shift[c] = fmaf(-mean[c], sgm, beta[c]);
    }
}

hipError_t launch_bn_fold(const float *mean, const float *var, const float *gamma,
                          const float *beta, int C, float *scale, float *shift, hipStream_t s)
{
    hipLaunchKernelGGL(k_bn_fold, dim3(cdiv(C, 64)), dim3(64), 0, s, mean, var, gamma, beta, C,
                       scale, shift);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// tf.image.resize_bilinear, TF-1.13 defaults: align_corners=False, legacy mapping
// src = dst * (in/out) (no half-pixel offset); lerp in fp32:  top + (bottom - top) * y_lerp, where
// top = tl + (tr - tl) * x_lerp   (inference.py:96-99).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_bilinear(const float *__restrict__ x, int N, int H,
                                                         int W, int C, int OH, int OW,
                                                         float hs, float ws, float *__restrict__ y)
{
    const long total = (long)N * OH * OW * C;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C);
        const long pix = o / C;
        const int ox = (int)(pix % OW);
        const int oy = (int)((pix / OW) % OH);
        const int n = (int)(pix / ((long)OW * OH));
        const float fy = (float)oy * hs, fx = (float)ox * ws;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float *img = x + (long)n * H * W * C;
        const float tl = img[((long)y0 * W + x0) * C + c], tr = img[((long)y0 * W + x1) * C + c];
        const float bl = img[((long)y1 * W + x0) * C + c], br = img[((long)y1 * W + x1) * C + c];
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        y[o] = top + (bot - top) * ly;
    }
}

// the same on channel quads (C % 4 == 0, 16-byte aligned tensors): one 16-byte access per neighbour instead of four 4-byte ones
__global__ __launch_bounds__(256) void k_resize_bilinear4(const float4 *__restrict__ x, int N, int H, int W, int C4, int OH,
                                                          int OW, float hs, float ws, float4 *__restrict__ y)
{
    const long total = (long)N * OH * OW * C4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C4);
        const long pix = o / C4;
        const int ox = (int)(pix % OW);
        const int oy = (int)((pix / OW) % OH);
        const int n = (int)(pix / ((long)OW * OH));
        const float fy = (float)oy * hs, fx = (float)ox * ws;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float4 *img = x + (long)n * H * W * C4;
        const float4 tl = img[((long)y0 * W + x0) * C4 + c], tr = img[((long)y0 * W + x1) * C4 + c];
        const float4 bl = img[((long)y1 * W + x0) * C4 + c], br = img[((long)y1 * W + x1) * C4 + c];
        auto l1 = [&](float ctl, float ctr, float cbl, float cbr) {
            const float top = ctl + (ctr - ctl) * lx;
            const float bot = cbl + (cbr - cbl) * lx;
            return top + (bot - top) * ly;
        };
        y[o] = make_float4(l1(tl.x, tr.x, bl.x, br.x), l1(tl.y, tr.y, bl.y, br.y), l1(tl.z, tr.z, bl.z, br.z), l1(tl.w, tr.w, bl.w, br.w));
    }
}

hipError_t launch_resize_bilinear(const float *x, int N, int H, int W, int C, int OH, int OW,
                                  float *y, hipStream_t s)
{
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const long total4 = (long)N * OH * OW * (C / 4);
        int grid4 = cdiv(total4, 256);
        if (grid4 > 65536) grid4 = 65536;
        hipLaunchKernelGGL(k_resize_bilinear4, dim3(grid4), dim3(256), 0, s, (const float4 *)x, N, H, W, C / 4, OH, OW,
                           (float)H / (float)OH, (float)W / (float)OW, (float4 *)y);
        return hipGetLastError();
    }
    const long total = (long)N * OH * OW * C;
    int grid = cdiv(total, 256);
    if (grid > 65536) grid = 65536;
    const float hs = (float)H / (float)OH, ws = (float)W / (float)OW;
    hipLaunchKernelGGL(k_resize_bilinear, dim3(grid), dim3(256), 0, s, x, N, H, W, C, OH, OW, hs,
                       ws, y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Synthetic frames: counter-based, so frame f is a pure function of (seed, f) on host
// and device alike (host twin: synthetic.py).  coarse 8x8 blocks + fine noise, per-frame brightness.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename TO>
__global__ __launch_bounds__(256) void k_synth_frames(uint64_t seed, long first, int count, int H,
                                                      int W, int C, TO *__restrict__ out)
{
    const long per = (long)H * W * C;
    const long total = per * count;
    const int Wc = W / 8;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long f = first + o / per;
        const long r = o % per;
        const int c = (int)(r % C);
        const long pix = r / C;
        const int xx = (int)(pix % W), yy = (int)(pix / W);
        const uint64_t fk = splitmix64(seed ^ ((uint64_t)f * 0xD1342543DE82EF95ull));
        const float u = (float)(splitmix64(fk ^ 0xB5ull) >> 40) * (1.0f / 16777216.0f);
        const float bright = 0.2f + 0.8f * u;
        const uint64_t ic = (uint64_t)(((long)(yy / 8) * Wc + (xx / 8)) * C + c);
        const int coarse = (int)(splitmix64(fk + 2ull * ic) >> 56);
        const int fine = (int)((splitmix64(fk + 2ull * (uint64_t)r + 1ull) >> 32) % 33ull) - 16;
        float v = (float)(coarse + fine) * bright;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        const int u8 = (int)v;
        if (sizeof(TO) == 1) out[o] = (TO)u8;                          // the decoded frame
        else out[o] = (TO)((float)u8 * (1.0f / 255.0f));                // after convert_image_dtype
    }
}

hipError_t launch_synth_frames(uint64_t seed, int64_t first, int count, int H, int W, int C,
                               void *out, bool out_is_u8, hipStream_t s)
{
    const long total = (long)H * W * C * count;
    int grid = cdiv(total, 256);
    if (grid > 262144) grid = 262144;
    if (out_is_u8)
        hipLaunchKernelGGL(k_synth_frames<uint8_t>, dim3(grid), dim3(256), 0, s, seed, (long)first, count, H, W, C,
                           (uint8_t *)out);
    else
        hipLaunchKernelGGL(k_synth_frames<float>, dim3(grid), dim3(256), 0, s, seed, (long)first, count, H, W, C,
                           (float *)out);
    return hipGetLastError();
}

}  // namespace ssal
