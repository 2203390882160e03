// ssal_bottleneck_mfma16.hip -- the narrow ENet blocks (bottleneck width <= 16) fused on
// v_mfma_f32_16x16x4_f32:  regular bottlenecks of stage 1/4 (C=64, F=16) and stage 5 (C=16, F=4),
// the 64 -> 16 upsample block (Bottleneck5_0) and the 16 -> 64 downsample block (Bottleneck1_0).
// Same structure, same accumulation order (bit-identical results) as the 32-wide family in
// ssal_bottleneck_mfma.hip; these blocks sit at 1/2 and 1/4 resolution with 16-64 channels and are
// HBM-bound, so the point of the fusion is one read of the input and one write of the output.
//
// lane maps (l = lane, i = l & 15, g = l >> 4):  A[row i][k = g],  B[k = g][col i],
// D reg r = D[row 4g + r][col i].  An instruction consumes k = 0..3 in lane-quarter order, so quarter
// g feeds channel 4s + g at step s; transpose4() produces that from per-lane consecutive channels.
// Bottleneck widths below 16 use zero-padded weight columns (exact: adds +0 to unused rows).
#include "ssal_internal.h"
#include "ssal_mfma.h"
#include "ssal_bottleneck_args.h"
#include "ssal_prof.h"
#include <type_traits>

namespace ssal {

constexpr int PMAX16 = 352;  // >= (8+2)*(32+2), multiple of 16

// ---- weights of the 3x3 conv FF -> FF as A-operand fragments, loaded ONCE per wave (they are
// identical for every M-tile): wcr[tap*KF + s] = Wc[tap][ci = 4s + g][co = i16] (0 beyond FF) ------
template <int FF, typename Args>
__device__ __forceinline__ void load_conv16_weights(const Args &a, int i16, int g, float (&wcr)[9 * (FF / 4)])
{
    constexpr int KF = FF / 4;
    const bool cval = i16 < FF;
    const int ic = cval ? i16 : 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int s = 0; s < KF; ++s) {
            const float w = a.wc[((tap * FF) + 4 * s + g) * FF + ic];
            wcr[tap * KF + s] = cval ? w : 0.0f;
        }
}

// ---- 3x3 conv FF -> FF over the LDS tile for one 16-pixel M-tile, + BN + PReLU; returns the result
// in GEMM-operand form: q[s] of lane (pixel i16, quarter g) = Q[pixel][ci = 4s + g] ------------------
template <int TW, int FF>
__device__ __forceinline__ void conv16_tile_q(const float *P, const float (&wcr)[9 * (FF / 4)],
                                              const float (&cs)[4], const float (&ct)[4],
                                              const float (&ca)[4], int mt, int i16, int g,
                                              float (&q)[4])
{
    constexpr int PS = FF + 2, HW2 = TW + 2, KF = FF / 4;
    const int t = mt * 16 + i16;
    const int r_ = t / TW, c_ = t - r_ * TW;
    f32x4 acc = {0};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const float *pq = P + ((r_ + kh) * HW2 + (c_ + kw)) * PS + g;
#pragma unroll
        for (int s = 0; s < KF; ++s)  // ci = 4s + g: ascending across the lane quarters
            acc = mfma16(wcr[tap * KF + s], pq[4 * s], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)  // row 4g + r = output channel; rows >= FF carry cs = ct = 0 -> exact 0
        q[r] = prelu1(fmaf(acc[r], cs[r], ct[r]), ca[r]);
    transpose4(q[0], q[1], q[2], q[3]);
}

// BN + PReLU constants of the conv for the rows (channels 4g + r) this lane holds; 0 beyond FF
template <int FF, typename Args>
__device__ __forceinline__ void load_conv16_bn(const Args &a, int g, float (&cs)[4], float (&ct)[4],
                                               float (&ca)[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * g + r;
        const bool ok = co < FF;
        const int cc = ok ? co : 0;
        const float s_ = a.cs[cc], t_ = a.ct[cc], a_ = a.ca[cc];
        cs[r] = ok ? s_ : 0.0f;
        ct[r] = ok ? t_ : 0.0f;
        ca[r] = ok ? a_ : 0.0f;
    }
}

// =================================================================================================
// regular / dilated bottleneck, CC channels, width FF = CC/4   (Bottleneck.call, enet_modules.py:526-599)
// The expansion GEMM is evaluated as D[co][pixel] = We^T[co][ci] * Q[ci][pixel]: lane = pixel and the
// 4 registers are 4 CONSECUTIVE output channels, so the residual and the output are one float4 per
// lane and N-tile.  The residual is never re-read: phase A projects the tile's CENTRE pixels with the
// same (wave, M-tile, lane) mapping phase B uses, and the float4 activation fragments it loads
// (channels 16m + 4g .. +3 of the lane's pixel) are exactly the residual fragments of N-tile m, so they
// stay in registers across the barrier; the one-pixel halo ring is projected separately.
// =================================================================================================
template <int TW, int CC, int FF>
__global__ __launch_bounds__(256, 3) void k_bottleneck16(BnkArgs a)
{
    constexpr int PS = FF + 2, KF = FF / 4, NT = CC / 16, HW2 = TW + 2, KP = CC / 4;
    constexpr int TH = 8;                      // tile rows (launcher guarantees a.TH == 8)
    constexpr int MPW = (TH * TW) / 16 / 4;    // centre M-tiles per wave: 4 (TW 32) or 2 (TW 16)
    constexpr int RING = 2 * HW2 + 2 * TH;     // halo ring pixels: 84 or 52
    __shared__ float P[PMAX16 * PS];
    __shared__ float BNV[3 * CC];  // es | et | ra: read in phase B through LDS (no global load after a store)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int d = a.dil;
    if (threadIdx.x < 3 * CC / 4) {
        const int arr = threadIdx.x / (CC / 4), k4 = threadIdx.x % (CC / 4);
        const float *src = arr == 0 ? a.es : arr == 1 ? a.et : a.ra;
        reinterpret_cast<float4 *>(BNV)[threadIdx.x] = reinterpret_cast<const float4 *>(src)[k4];
    }
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int px = b % d; b /= d;
    const int py = b % d; b /= d;
    const int n = b;
    const int Hp = (a.H - py + d - 1) / d;
    const int Wp = (a.W - px + d - 1) / d;
    const int ty0 = ty * TH, tx0 = tx * TW;
    if (ty0 >= Hp || tx0 >= Wp) return;
    const float *ximg = a.x + (long)n * a.H * a.W * CC;
    float *yimg = a.y + (long)n * a.H * a.W * CC;
    PhaseTrace tr;
    tr.mark(0);

    // ---- phase A: projection of centre + ring into LDS -------------------------------------------
    const bool cval = i16 < FF;
    const int ic = cval ? i16 : 0;
    float wpr[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s) {
        const float w = a.wp[(4 * s + g) * FF + ic];
        wpr[s] = cval ? w : 0.0f;
    }
    const float bs = a.ps[ic], bt = a.pt[ic], ba = a.pa[ic];

    // halo'd-tile index q of (a) centre pixel t, (b) ring pixel u
    auto q_center = [&](int t) { return (t / TW + 1) * HW2 + (t % TW) + 1; };
    auto q_ring = [&](int u) {
        if (u < HW2) return u;
        if (u < 2 * HW2) return (TH + 1) * HW2 + (u - HW2);
        const int k = u - 2 * HW2;
        return (1 + (k >> 1)) * HW2 + ((k & 1) ? HW2 - 1 : 0);
    };
    // project 16 pixels (one per i16; q < 0 = padding lane) and write the 16 results to P
    auto project = [&](const float4 (&v)[NT], unsigned vmask, const int (&qrow)[4]) {
        f32x4 acc = {0};
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            float r0 = v[m].x, r1 = v[m].y, r2 = v[m].z, r3 = v[m].w;
            transpose4(r0, r1, r2, r3);  // reg r of quarter g: channel 16m + 4r + g
            acc = mfma16(r0, wpr[4 * m + 0], acc);
            acc = mfma16(r1, wpr[4 * m + 1], acc);
            acc = mfma16(r2, wpr[4 * m + 2], acc);
            acc = mfma16(r3, wpr[4 * m + 3], acc);
        }
        if (cval) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (vmask >> (4 * g + r)) & 1u;
                if (qrow[r] >= 0) P[qrow[r] * PS + i16] = ok ? prelu1(fmaf(acc[r], bs, bt), ba) : 0.0f;
            }
        }
    };

    float4 xk[MPW][NT];  // centre activation fragments == residual fragments of phase B
    int offk[MPW];       // element offset of the lane's pixel (+ 4g), -1 outside the image
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        const int t = mt * 16 + i16;
        const int pr = ty0 + t / TW, pc = tx0 + t % TW;
        const bool valid = (pr < Hp) && (pc < Wp);
        offk[k] = valid ? ((py + pr * d) * a.W + (px + pc * d)) * CC + 4 * g : -1;
        const float *xp = ximg + (valid ? offk[k] : 4 * g);
#pragma unroll
        for (int m = 0; m < NT; ++m) xk[k][m] = *reinterpret_cast<const float4 *>(xp + 16 * m);
    }
    // the halo-ring fragments are requested in the same breath (ONE exposed HBM latency per workgroup,
    // not two): this wave's ring M-tiles are wave, wave + 4 (RM = 2 for the 84-pixel ring of a 32-wide tile)
    constexpr int RM = (RING + 63) / 64;
    float4 xr[RM][NT];
    bool rvalid[RM];
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        const int u = (wave + 4 * k) * 16 + i16;
        const int q = u < RING ? q_ring(u) : 0;
        const int hr = q / HW2, hc = q - hr * HW2;
        const int pr = ty0 - 1 + hr, pc = tx0 - 1 + hc;
        rvalid[k] = (u < RING) && (pr >= 0) && (pr < Hp) && (pc >= 0) && (pc < Wp);
        const float *xp = rvalid[k] ? ximg + ((long)(py + pr * d) * a.W + (px + pc * d)) * CC + 4 * g : ximg + 4 * g;
#pragma unroll
        for (int m = 0; m < NT; ++m) xr[k][m] = *reinterpret_cast<const float4 *>(xp + 16 * m);
    }
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        const unsigned vmask = (unsigned)(__ballot(offk[k] >= 0) & 0xFFFFull);
        int qrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) qrow[r] = q_center(mt * 16 + 4 * g + r);
        if (!SSAL_ABLATE_IS(a, 2)) project(xk[k], vmask, qrow);
    }
    tr.mark(1);  // centre projected (activation loads have arrived)
#pragma unroll
    for (int k = 0; k < RM; ++k) {  // halo ring
        const int mtr = wave + 4 * k;
        if (mtr * 16 < RING) {  // wave-uniform
            const unsigned vmask = (unsigned)(__ballot(rvalid[k]) & 0xFFFFull);
            int qrow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ur = mtr * 16 + 4 * g + r;
                qrow[r] = ur < RING ? q_ring(ur) : -1;
            }
            if (!SSAL_ABLATE_IS(a, 2)) project(xr[k], vmask, qrow);
        }
    }

    // every loop-invariant operand of phase B is requested before the barrier: nothing in phase B waits
    // on a global load, so its stores never sit in front of a load in the (in-order) vmcnt queue.
    float wcr[9 * KF];
    load_conv16_weights<FF>(a, i16, g, wcr);
    float cs[4], ct[4], ca[4];
    load_conv16_bn<FF>(a, g, cs, ct, ca);
    float wer[NT * KF];  // We^T as A operand: row = co_local (i16), k = ci = 4s + g
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KF; ++s) wer[nt * KF + s] = a.we[(4 * s + g) * CC + nt * 16 + i16];

    tr.mark(2);  // ring projected, phase-B operands requested
    __syncthreads();
    tr.mark(3);
    if (SSAL_ABLATE_IS(a, 1)) return;

    // ---- phase B: conv, expansion, residual (from registers), store -------------------------------
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        float q[4];
        conv16_tile_q<TW, FF>(P, wcr, cs, ct, ca, mt, i16, g, q);
        if (k == 0) tr.mark(4);  // first conv done
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 e = {0};
#pragma unroll
            for (int s = 0; s < KF; ++s) e = mfma16(wer[nt * KF + s], q[s], e);
            // expansion BN / residual PReLU of channels nt*16 + 4g .. +3 (reg r = channel nt*16 + 4g + r)
            const float4 s1 = *reinterpret_cast<const float4 *>(BNV + nt * 16 + 4 * g);
            const float4 t1 = *reinterpret_cast<const float4 *>(BNV + CC + nt * 16 + 4 * g);
            const float4 al = *reinterpret_cast<const float4 *>(BNV + 2 * CC + nt * 16 + 4 * g);
            float4 o;
            o.x = prelu1(fmaf(e[0], s1.x, t1.x) + xk[k][nt].x, al.x);
            o.y = prelu1(fmaf(e[1], s1.y, t1.y) + xk[k][nt].y, al.y);
            o.z = prelu1(fmaf(e[2], s1.z, t1.z) + xk[k][nt].z, al.z);
            o.w = prelu1(fmaf(e[3], s1.w, t1.w) + xk[k][nt].w, al.w);
            if (offk[k] >= 0) *reinterpret_cast<float4 *>(yimg + offk[k] + nt * 16) = o;
        }
        if (k == 0) tr.mark(5);  // first M-tile stored (issued)
    }
    tr.mark(6);
#ifdef SSAL_PHASE_TRACE
    __builtin_amdgcn_s_waitcnt(0);  // mark 7 = all stores acknowledged
#endif
    tr.mark(7);
    tr.flush(a.trace, lane, wave);
}

// =================================================================================================
// downsample bottleneck 16 -> 64, width 8 (Bottleneck1_0; enet_modules.py:868-938)
// =================================================================================================
// The tile's centre output pixels are projected with the (wave, M-tile, lane) mapping phase B uses: the four
// 2x2-patch fragments the projection loads (channels 4g..4g+3 of each tap) are the pooling window of the
// residual, so the first-max pooling and its packed window codes are evaluated right there (codes stored at
// once, one pooled float4 per M-tile kept) and the block input is read once.  All patch loads of a wave
// (centre + its share of the halo ring) are requested up front.
template <int TW>
__global__ __launch_bounds__(256, 3) void k_downsample16(DownArgs a)
{
    constexpr int CI = 16, FF = 8, CO = 64, PS = FF + 2, HW2 = TW + 2;
    constexpr int TH = 8;                      // tile rows (launcher guarantees a.TH == 8)
    constexpr int MPW = (TH * TW) / 16 / 4;    // centre M-tiles per wave: 4 (TW 32) or 2 (TW 16)
    constexpr int RING = 2 * HW2 + 2 * TH;     // halo ring pixels: 84 or 52
    constexpr int RM = (RING + 63) / 64;       // ring M-tiles per wave (upper bound)
    __shared__ float P[PMAX16 * PS];
    __shared__ float BNV[3 * CO];  // es | et | ra, read in phase B through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    if (threadIdx.x < 3 * CO / 4) {
        const int arr = threadIdx.x / (CO / 4), k4 = threadIdx.x % (CO / 4);
        const float *src = arr == 0 ? a.es : arr == 1 ? a.et : a.ra;
        reinterpret_cast<float4 *>(BNV)[threadIdx.x] = reinterpret_cast<const float4 *>(src)[k4];
    }
    const int Ho = a.H / 2, Wo = a.W / 2;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int ty0 = ty * TH, tx0 = tx * TW;
    const float *ximg = a.x + (long)n * a.H * a.W * CI;
    float *yimg = a.y + (long)n * Ho * Wo * CO;
    uint8_t *cimg = a.code + (long)n * Ho * Wo * CI;
    const bool cval = i16 < FF;
    const int ic = cval ? i16 : 0;

    // ---- phase A: 2x2/s2 projection (K = 4 taps x 16 ci) of centre + ring -> LDS --------------------
    float wpr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // step s = 4*tap + s': ci = 4s' + g of tap (dy,dx)
        const float w = a.wp[((s >> 2) * CI + 4 * (s & 3) + g) * FF + ic];
        wpr[s] = cval ? w : 0.0f;
    }
    const float bs = a.ps[ic], bt = a.pt[ic], ba = a.pa[ic];
    auto q_center = [&](int t) { return (t / TW + 1) * HW2 + (t % TW) + 1; };
    auto q_ring = [&](int u) {
        if (u < HW2) return u;
        if (u < 2 * HW2) return (TH + 1) * HW2 + (u - HW2);
        const int k = u - 2 * HW2;
        return (1 + (k >> 1)) * HW2 + ((k & 1) ? HW2 - 1 : 0);
    };
    auto load_patch = [&](const float *xp, float4 (&v)[4]) {  // taps in (dy,dx) order = (kh,kw) order of the oracle
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
            v[tap] = *reinterpret_cast<const float4 *>(xp + ((tap >> 1) * a.W + (tap & 1)) * CI + 4 * g);
    };
    auto project = [&](const float4 (&v)[4], unsigned vmask, const int (&qrow)[4]) {
        f32x4 acc = {0};
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            float r0 = v[tap].x, r1 = v[tap].y, r2 = v[tap].z, r3 = v[tap].w;
            transpose4(r0, r1, r2, r3);
            acc = mfma16(r0, wpr[4 * tap + 0], acc);
            acc = mfma16(r1, wpr[4 * tap + 1], acc);
            acc = mfma16(r2, wpr[4 * tap + 2], acc);
            acc = mfma16(r3, wpr[4 * tap + 3], acc);
        }
        if (cval) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (vmask >> (4 * g + r)) & 1u;
                if (qrow[r] >= 0) P[qrow[r] * PS + i16] = ok ? prelu1(fmaf(acc[r], bs, bt), ba) : 0.0f;
            }
        }
    };

    float4 vk[MPW][4], vr[RM][4];
    long opixk[MPW];  // output pixel index of the lane's centre pixel, -1 outside the image
    bool rvalid[RM];
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int t = (wave + 4 * k) * 16 + i16;
        const int oy = ty0 + t / TW, ox = tx0 + t % TW;
        const bool valid = (oy < Ho) && (ox < Wo);
        opixk[k] = valid ? (long)oy * Wo + ox : -1;
        load_patch(valid ? ximg + ((long)(2 * oy) * a.W + 2 * ox) * CI : ximg, vk[k]);
    }
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        const int u = (wave + 4 * k) * 16 + i16;
        const int q = u < RING ? q_ring(u) : 0;
        const int pr = ty0 - 1 + q / HW2, pc = tx0 - 1 + q % HW2;
        rvalid[k] = (u < RING) && (pr >= 0) && (pr < Ho) && (pc >= 0) && (pc < Wo);
        load_patch(rvalid[k] ? ximg + ((long)(2 * pr) * a.W + 2 * pc) * CI : ximg, vr[k]);
    }
    __builtin_amdgcn_sched_barrier(0);

    float4 pooled[MPW];  // max-pooled input, channels 4g..4g+3 of the lane's centre pixel (N-tile 0 residual)
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        const unsigned vmask = (unsigned)(__ballot(opixk[k] >= 0) & 0xFFFFull);
        int qrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) qrow[r] = q_center(mt * 16 + 4 * g + r);
        project(vk[k], vmask, qrow);
        // 2x2 window, strict '>' scan in (dy,dx) order: the first maximum wins (TF's rule)
        const float c00[4] = {vk[k][0].x, vk[k][0].y, vk[k][0].z, vk[k][0].w};
        const float c01[4] = {vk[k][1].x, vk[k][1].y, vk[k][1].z, vk[k][1].w};
        const float c10[4] = {vk[k][2].x, vk[k][2].y, vk[k][2].z, vk[k][2].w};
        const float c11[4] = {vk[k][3].x, vk[k][3].y, vk[k][3].z, vk[k][3].w};
        float best[4];
        unsigned packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float bv = c00[r];
            unsigned cd = 0;
            if (c01[r] > bv) { bv = c01[r]; cd = 1; }
            if (c10[r] > bv) { bv = c10[r]; cd = 2; }
            if (c11[r] > bv) { bv = c11[r]; cd = 3; }
            best[r] = bv;
            packed |= cd << (8 * r);
        }
        pooled[k] = make_float4(best[0], best[1], best[2], best[3]);
        if (opixk[k] >= 0) *reinterpret_cast<unsigned *>(cimg + opixk[k] * CI + 4 * g) = packed;  // channels 4g..4g+3
    }
#pragma unroll
    for (int k = 0; k < RM; ++k) {  // halo ring
        const int mtr = wave + 4 * k;
        if (mtr * 16 < RING) {  // wave-uniform
            const unsigned vmask = (unsigned)(__ballot(rvalid[k]) & 0xFFFFull);
            int qrow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ur = mtr * 16 + 4 * g + r;
                qrow[r] = ur < RING ? q_ring(ur) : -1;
            }
            project(vr[k], vmask, qrow);
        }
    }

    // loop-invariant operands of phase B, requested before the barrier
    float wcr[9 * (FF / 4)];
    load_conv16_weights<FF>(a, i16, g, wcr);
    float cs[4], ct[4], ca[4];
    load_conv16_bn<FF>(a, g, cs, ct, ca);
    float wer[(CO / 16) * 2];  // We^T as A operand: row = co_local (i16), k = ci = 4s + g, s < 2
#pragma unroll
    for (int nt = 0; nt < CO / 16; ++nt) {
        wer[nt * 2 + 0] = a.we[(0 + g) * CO + nt * 16 + i16];
        wer[nt * 2 + 1] = a.we[(4 + g) * CO + nt * 16 + i16];
    }
    __syncthreads();

    // ---- phase B: 3x3 conv (8 -> 8), expansion (8 -> 64) as D[co][pixel] (lane = pixel, 4 consecutive
    // channels per lane -> float4 stores), + pooled residual on N-tile 0; no global load in this phase ----
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        float q[4];
        conv16_tile_q<TW, FF>(P, wcr, cs, ct, ca, mt, i16, g, q);
#pragma unroll
        for (int nt = 0; nt < CO / 16; ++nt) {
            f32x4 e = {0};
            e = mfma16(wer[nt * 2 + 0], q[0], e);
            e = mfma16(wer[nt * 2 + 1], q[1], e);
            const int co = nt * 16 + 4 * g;  // reg r = channel co + r
            const float4 s1 = *reinterpret_cast<const float4 *>(BNV + co);
            const float4 t1 = *reinterpret_cast<const float4 *>(BNV + CO + co);
            const float4 al = *reinterpret_cast<const float4 *>(BNV + 2 * CO + co);
            const float4 rr4 = nt == 0 ? pooled[k] : make_float4(0.f, 0.f, 0.f, 0.f);  // channels >= 16: zero padding
            float4 o;
            o.x = prelu1(fmaf(e[0], s1.x, t1.x) + rr4.x, al.x);
            o.y = prelu1(fmaf(e[1], s1.y, t1.y) + rr4.y, al.y);
            o.z = prelu1(fmaf(e[2], s1.z, t1.z) + rr4.z, al.z);
            o.w = prelu1(fmaf(e[3], s1.w, t1.w) + rr4.w, al.w);
            if (opixk[k] >= 0) *reinterpret_cast<float4 *>(yimg + opixk[k] * CO + co) = o;
        }
    }
}

// =================================================================================================
// Initial block + Bottleneck1_0 in ONE launch (enet_modules.py:190-224 + 868-938).
// Unfused, the Initial kernel writes its [N,H/2,W/2,16] output (268 MB at batch 8 x 1024 x 2048) and k_downsample16
// reads it back at once: both launches are HBM-bound.  Here a workgroup owns an 8 x 16 tile of Bottleneck1_0's OUTPUT:
//   phase 0  the 41 x 73 image window behind the halo'd tile -> LDS (coalesced rows; uint8 frames converted on the way,
//            x * (1/255) as tf.image.convert_image_dtype); the expansion BN vectors -> LDS
//   phase A  per 16-pixel M-tile, lane (pixel, quarter g) EVALUATES the four 2x2-patch fragments the projection needs --
//            Initial's output channels 4g..4g+3 at the four patch positions: conv 3x3/s2 on the matrix cores, one exact fmaf
//            chain per channel over (kh, kw, ci) ascending (the order of k_initial and of the oracle; window taps outside
//            the image are exact zeros), 2x2 max-pool of the image for the concatenated channels, BN + PReLU -- instead of
//            loading them;
//            from there on the kernel IS k_downsample16: projection GEMM (K = 4 taps x 16), first-max pooling of the
//            patch + packed window codes, halo ring, P in LDS
//   phase B  unchanged: 3x3 conv (8 -> 8), expansion (8 -> 64) + pooled residual + PReLU, float4 stores.
// Initial's output never exists (it is no endpoint: enet.py:311-318); the per-layer entry points keep the two kernels.
// The halo recompute of Initial is 720 / 512 = 1.41x of a 351-FMA-per-pixel convolution (7 MFMAs per 16 pixels).
// =================================================================================================
struct InitDownArgs {
    const void *img;                                // [N,H,W,CIN] float32 in [0,1] or the decoded uint8 frame
    const float *iw, *iscale, *ishift, *ialpha;     // Initial: kernel [3][3][CIN][16-CIN], folded BN [16], alpha [16]
    DownArgs d;                                     // Bottleneck1_0: d.x unused, d.H / d.W = Initial's output dims
};

__device__ __forceinline__ float unit_of(float v) { return v; }
__device__ __forceinline__ float unit_of(uint8_t v) { return (float)v * (1.0f / 255.0f); }

template <int CIN, typename TX>
__global__ __launch_bounds__(256, 3) void k_initial_down16(InitDownArgs A)
{
    constexpr int TW = 16, TH = 8, CI = 16, FF = 8, CO = 64, PS = FF + 2, HW2 = TW + 2;
    constexpr int CC = 16 - CIN;               // convolution channels of Initial; channels CC..15 = pooled image
    constexpr int MPW = (TH * TW) / 16 / 4;    // centre M-tiles per wave: 2
    constexpr int RING = 2 * HW2 + 2 * TH;     // halo ring pixels: 52 = 4 M-tiles, one per wave
    constexpr int WR = 4 * (TH + 2) + 1, WC = 4 * (TW + 2) + 1, WCF = WC * CIN;  // image window: 41 rows x 73 pixels
    constexpr int PROWS = 192;                 // >= (TH+2)*(TW+2) = 180
    __shared__ float IMG[WR * WCF];
    __shared__ float P[PROWS * PS];
    __shared__ float BNV[3 * CO];
    const DownArgs &a = A.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int Ho = a.H / 2, Wo = a.W / 2, HI = 2 * a.H, WIM = 2 * a.W;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int ty0 = ty * TH, tx0 = tx * TW;
    float *yimg = a.y + (long)n * Ho * Wo * CO;
    uint8_t *cimg = a.code + (long)n * Ho * Wo * CI;

    // ---- phase 0 ---------------------------------------------------------------------------------------
    {
        const TX *ximg = reinterpret_cast<const TX *>(A.img) + (long)n * HI * WIM * CIN;
        const int r0 = 4 * ty0 - 4, cf0 = (4 * tx0 - 4) * CIN;
        // every load of the thread is requested before the first one is used (a rolled load -> store loop exposes the
        // memory latency once per element: 36 round trips per workgroup); (row, offset) advance by constants
        // Window rows start on a 4-element boundary of the image row (column offset (4 tx0 - 4) CIN elements, row pitch W CIN
        // elements with W % 4 == 0), so the window is fetched in QUADS of four elements -- one 16-byte load per quad of a
        // float32 frame, one 4-byte load per quad of a uint8 frame: 9 instead of 36 loads per thread (round 5: 228 -> 195 us).
        // A quad lies wholly inside or wholly outside the image row; the last quad of a window row reaches up to 3
        // elements beyond the window (CIN = 3: one): loaded, not stored.  Every load is requested before the first is used.
        constexpr int QPR = (WCF + 3) / 4, NQ = WR * QPR, NITQ = (NQ + 255) / 256;
        float4 tq[NITQ];
        int wr = (int)threadIdx.x / QPR, wq_ = (int)threadIdx.x % QPR;
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
            const int gy = r0 + wr, gf = cf0 + 4 * wq_;
            const bool ok = wr < WR && gy >= 0 && gy < HI && gf >= 0 && gf + 3 < WIM * CIN;
            const TX *src = ximg + (ok ? (long)gy * WIM * CIN + gf : 0);
            if (std::is_same<TX, float>::value) {
                tq[it] = *reinterpret_cast<const float4 *>(src);
            } else {
                const unsigned u = *reinterpret_cast<const unsigned *>(src);
                tq[it] = make_float4(unit_of((uint8_t)(u & 0xFFu)), unit_of((uint8_t)((u >> 8) & 0xFFu)),
                                     unit_of((uint8_t)((u >> 16) & 0xFFu)), unit_of((uint8_t)(u >> 24)));
            }
            if (!ok) tq[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            wr += 256 / QPR; wq_ += 256 % QPR;
            if (wq_ >= QPR) { wq_ -= QPR; ++wr; }
        }
        wr = (int)threadIdx.x / QPR; wq_ = (int)threadIdx.x % QPR;
#pragma unroll
        for (int it = 0; it < NITQ; ++it) {
            if (wr < WR) {
                float *dst = IMG + wr * WCF + 4 * wq_;
                const float v[4] = {tq[it].x, tq[it].y, tq[it].z, tq[it].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * wq_ + e < WCF) dst[e] = v[e];
            }
            wr += 256 / QPR; wq_ += 256 % QPR;
            if (wq_ >= QPR) { wq_ -= QPR; ++wr; }
        }
        if (threadIdx.x < 3 * CO / 4) {
            const int arr = threadIdx.x / (CO / 4), k4 = threadIdx.x % (CO / 4);
            const float *src = arr == 0 ? a.es : arr == 1 ? a.et : a.ra;
            reinterpret_cast<float4 *>(BNV)[threadIdx.x] = reinterpret_cast<const float4 *>(src)[k4];
        }
    }
    const bool cval = i16 < FF;
    const int ic = cval ? i16 : 0;
    float wpr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // step s = 4*tap + s': ci = 4s' + g of tap (dy,dx)
        const float w = a.wp[((s >> 2) * CI + 4 * (s & 3) + g) * FF + ic];
        wpr[s] = cval ? w : 0.0f;
    }
    const float bs = a.ps[ic], bt = a.pt[ic], ba = a.pa[ic];
    const float4 isc = *reinterpret_cast<const float4 *>(A.iscale + 4 * g);
    const float4 ish = *reinterpret_cast<const float4 *>(A.ishift + 4 * g);
    const float4 ial = *reinterpret_cast<const float4 *>(A.ialpha + 4 * g);
    __syncthreads();

    // ---- phase A ---------------------------------------------------------------------------------------
    auto q_center = [&](int t) { return (t / TW + 1) * HW2 + (t % TW) + 1; };
    auto q_ring = [&](int u) {
        if (u < HW2) return u;
        if (u < 2 * HW2) return (TH + 1) * HW2 + (u - HW2);
        const int k = u - 2 * HW2;
        return (1 + (k >> 1)) * HW2 + ((k & 1) ? HW2 - 1 : 0);
    };
    // Initial's convolution on the matrix cores (the fp32 MFMA rate equals the packed-FMA rate, but the operands need no
    // per-lane broadcast copies): D[co][pixel] = W^T[co][k] * X[k][pixel], k = (kh, kw, ci) ascending = the fmaf-chain
    // order of k_initial and of the oracle; K = 9*CIN padded to a multiple of 4 with zero kernel rows.  A operand: the
    // kernel, row co = i16 (zero rows for the pooled channels), k = 4s + g -- NS registers, loop-invariant.  B operand:
    // lane (pixel i16, k = 4s + g) reads ITS tap of the pixel's 3x3 window straight from the LDS image.
    constexpr int NS = (9 * CIN + 3) / 4;
    float wI[NS];
    int koff[NS];
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int k = 4 * s_ + g;
        const bool kv = k < 9 * CIN;
        const float w = A.iw[(kv ? k : 0) * CC + (i16 < CC ? i16 : 0)];
        wI[s_] = (kv && i16 < CC) ? w : 0.0f;
        koff[s_] = kv ? (k / (3 * CIN)) * WCF + (k % (3 * CIN)) : 0;  // (kw, ci) are contiguous inside a window row
    }
    // Initial's output channels 4g..4g+3 at the 2x2 patch of halo'd-tile pixel q (row hr, column hc): v[tap], tap = dy*2+dx
    auto eval_patch = [&](int q, float4 (&v)[4]) {
        const int hr = q / HW2, hc = q - hr * HW2;
        const float *wp0 = IMG + (4 * hr) * WCF + (4 * hc) * CIN;  // 5 x 5 pixel window of the image
        const float scv[4] = {isc.x, isc.y, isc.z, isc.w}, shv[4] = {ish.x, ish.y, ish.z, ish.w},
                    alv[4] = {ial.x, ial.y, ial.z, ial.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float *wt = wp0 + (2 * (t >> 1)) * WCF + (2 * (t & 1)) * CIN;  // window of Initial pixel (2y+dy, 2x+dx)
            f32x4 acc = {0};
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) acc = mfma16(wI[s_], wt[koff[s_]], acc);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float val = acc[r];
                if (12 + r >= CC) {  // channels >= CC (quarter 3 only): 2x2 max-pool of the image (k_initial scans the window
                    const int ci = 12 + r - CC;  // with a strict '>': the VALUE of the first maximum is the maximum)
                    const float bv = fmaxf(fmaxf(wt[ci], wt[CIN + ci]), fmaxf(wt[WCF + ci], wt[WCF + CIN + ci]));
                    val = g == 3 ? bv : val;
                }
                o[r] = prelu1(fmaf(val, scv[r], shv[r]), alv[r]);
            }
            v[t] = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    auto project = [&](const float4 (&v)[4], unsigned vmask, const int (&qrow)[4]) {
        f32x4 acc = {0};
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            float r0 = v[tap].x, r1 = v[tap].y, r2 = v[tap].z, r3 = v[tap].w;
            transpose4(r0, r1, r2, r3);
            acc = mfma16(r0, wpr[4 * tap + 0], acc);
            acc = mfma16(r1, wpr[4 * tap + 1], acc);
            acc = mfma16(r2, wpr[4 * tap + 2], acc);
            acc = mfma16(r3, wpr[4 * tap + 3], acc);
        }
        if (cval) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (vmask >> (4 * g + r)) & 1u;
                if (qrow[r] >= 0) P[qrow[r] * PS + i16] = ok ? prelu1(fmaf(acc[r], bs, bt), ba) : 0.0f;
            }
        }
    };

    float4 pooled[MPW];
    long opixk[MPW];
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        const int t = mt * 16 + i16;
        const int oy = ty0 + t / TW, ox = tx0 + t % TW;
        const bool valid = (oy < Ho) && (ox < Wo);
        opixk[k] = valid ? (long)oy * Wo + ox : -1;
        float4 vk[4];
        eval_patch(q_center(t), vk);
        const unsigned vmask = (unsigned)(__ballot(valid) & 0xFFFFull);
        int qrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) qrow[r] = q_center(mt * 16 + 4 * g + r);
        project(vk, vmask, qrow);
        // 2x2 window, strict '>' scan in (dy,dx) order: the first maximum wins (TF's rule)
        const float c00[4] = {vk[0].x, vk[0].y, vk[0].z, vk[0].w};
        const float c01[4] = {vk[1].x, vk[1].y, vk[1].z, vk[1].w};
        const float c10[4] = {vk[2].x, vk[2].y, vk[2].z, vk[2].w};
        const float c11[4] = {vk[3].x, vk[3].y, vk[3].z, vk[3].w};
        float best[4];
        unsigned packed = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float bv = c00[r];
            unsigned cd = 0;
            if (c01[r] > bv) { bv = c01[r]; cd = 1; }
            if (c10[r] > bv) { bv = c10[r]; cd = 2; }
            if (c11[r] > bv) { bv = c11[r]; cd = 3; }
            best[r] = bv;
            packed |= cd << (8 * r);
        }
        pooled[k] = make_float4(best[0], best[1], best[2], best[3]);
        if (valid) *reinterpret_cast<unsigned *>(cimg + opixk[k] * CI + 4 * g) = packed;  // channels 4g..4g+3
    }
    {  // halo ring: M-tile `wave` of the 52 ring pixels
        const int u = wave * 16 + i16;
        const int q = u < RING ? q_ring(u) : 0;
        const int pr = ty0 - 1 + q / HW2, pc = tx0 - 1 + q % HW2;
        const bool rvalid = (u < RING) && (pr >= 0) && (pr < Ho) && (pc >= 0) && (pc < Wo);
        float4 vr[4];
        eval_patch(q, vr);
        const unsigned vmask = (unsigned)(__ballot(rvalid) & 0xFFFFull);
        int qrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ur = wave * 16 + 4 * g + r;
            qrow[r] = ur < RING ? q_ring(ur) : -1;
        }
        project(vr, vmask, qrow);
    }

    // loop-invariant operands of phase B, requested before the barrier
    float wcr[9 * (FF / 4)];
    load_conv16_weights<FF>(a, i16, g, wcr);
    float cs[4], ct[4], ca[4];
    load_conv16_bn<FF>(a, g, cs, ct, ca);
    float wer[(CO / 16) * 2];
#pragma unroll
    for (int nt = 0; nt < CO / 16; ++nt) {
        wer[nt * 2 + 0] = a.we[(0 + g) * CO + nt * 16 + i16];
        wer[nt * 2 + 1] = a.we[(4 + g) * CO + nt * 16 + i16];
    }
    __syncthreads();

    // ---- phase B: as k_downsample16 ----------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int mt = wave + 4 * k;
        float q[4];
        conv16_tile_q<TW, FF>(P, wcr, cs, ct, ca, mt, i16, g, q);
#pragma unroll
        for (int nt = 0; nt < CO / 16; ++nt) {
            f32x4 e = {0};
            e = mfma16(wer[nt * 2 + 0], q[0], e);
            e = mfma16(wer[nt * 2 + 1], q[1], e);
            const int co = nt * 16 + 4 * g;  // reg r = channel co + r
            const float4 s1 = *reinterpret_cast<const float4 *>(BNV + co);
            const float4 t1 = *reinterpret_cast<const float4 *>(BNV + CO + co);
            const float4 al = *reinterpret_cast<const float4 *>(BNV + 2 * CO + co);
            const float4 rr4 = nt == 0 ? pooled[k] : make_float4(0.f, 0.f, 0.f, 0.f);  // channels >= 16: zero padding
            float4 o;
            o.x = prelu1(fmaf(e[0], s1.x, t1.x) + rr4.x, al.x);
            o.y = prelu1(fmaf(e[1], s1.y, t1.y) + rr4.y, al.y);
            o.z = prelu1(fmaf(e[2], s1.z, t1.z) + rr4.z, al.z);
            o.w = prelu1(fmaf(e[3], s1.w, t1.w) + rr4.w, al.w);
            if (opixk[k] >= 0) *reinterpret_cast<float4 *>(yimg + opixk[k] * CO + co) = o;
        }
    }
}

// =================================================================================================
// upsample bottleneck 64 -> 16 (Bottleneck5_0; enet_modules.py:1217-1292): proj 64 -> 16, transposed
// conv 16 -> 8 with two output-parity classes stacked in the 16 MFMA rows ([ee|eo] and [oe|oo], see
// ssal_bottleneck_mfma.hip), exp 8 -> 16, residual 1x1 conv 64 -> 16 + gather-unpool.
// =================================================================================================
// Centre pixels are projected with the (wave, M-tile, lane) mapping phase B uses and the same transposed
// activation registers feed the 1x1 residual conv (64 -> 16, D[co][pixel]) in the same pass: the block input
// is read once, the residual (4 registers per M-tile) and the packed window codes wait in registers.
template <int TW>
__global__ __launch_bounds__(256, 3) void k_upsample16(UpArgs a)
{
    constexpr int CI = 64, PF = 16, CF = 8, CO = 16, PS = PF + 2, HW2 = TW + 2, NT = CI / 16;
    constexpr int TH = 8;                      // tile rows (launcher guarantees a.TH == 8)
    constexpr int MPW = (TH * TW) / 16 / 4;    // centre M-tiles per wave: 4 (TW 32) or 2 (TW 16)
    constexpr int RING = 2 * HW2 + 2 * TH;     // halo ring pixels: 84 or 52
    constexpr int RM = (RING + 63) / 64;       // ring M-tiles per wave (upper bound)
    __shared__ float P[PMAX16 * PS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int n = b;
    const int ty0 = ty * TH, tx0 = tx * TW;
    const float *ximg = a.x + (long)n * a.H * a.W * CI;
    const uint8_t *cimg = a.code + (long)n * a.H * a.W * CO;
    float *yimg = a.y + (long)n * 4 * a.H * a.W * CO;

    // ---- phase A: projection 64 -> 16 (+ residual conv 64 -> 16 on the centre) ---------------------------
    float wpr[CI / 4], wrr[CI / 4];
#pragma unroll
    for (int s_ = 0; s_ < CI / 4; ++s_) {
        wpr[s_] = a.wp[(4 * s_ + g) * PF + i16];  // B operand of the projection: Wp[ci = 4s + g][co = i16]
        wrr[s_] = a.wr[(4 * s_ + g) * CO + i16];  // A operand of the residual conv: Wr^T[co = i16][ci = 4s + g]
    }
    const float bs = a.ps[i16], bt = a.pt[i16], ba = a.pa[i16];
    auto q_center = [&](int t) { return (t / TW + 1) * HW2 + (t % TW) + 1; };
    auto q_ring = [&](int u) {
        if (u < HW2) return u;
        if (u < 2 * HW2) return (TH + 1) * HW2 + (u - HW2);
        const int k = u - 2 * HW2;
        return (1 + (k >> 1)) * HW2 + ((k & 1) ? HW2 - 1 : 0);
    };
    auto load_frags = [&](const float *xp, float4 (&v)[NT]) {  // quarter g takes the g-th float4 of every 16 channels
#pragma unroll
        for (int m = 0; m < NT; ++m) v[m] = *reinterpret_cast<const float4 *>(xp + 16 * m + 4 * g);
    };
    auto project = [&](const float4 (&v)[NT], unsigned vmask, const int (&qrow)[4], bool with_res, f32x4 &res) {
        f32x4 acc = {0};
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            float r0 = v[m].x, r1 = v[m].y, r2 = v[m].z, r3 = v[m].w;
            transpose4(r0, r1, r2, r3);  // reg r of quarter g: channel 16m + 4r + g
            acc = mfma16(r0, wpr[4 * m + 0], acc);
            acc = mfma16(r1, wpr[4 * m + 1], acc);
            acc = mfma16(r2, wpr[4 * m + 2], acc);
            acc = mfma16(r3, wpr[4 * m + 3], acc);
            if (with_res) {
                res = mfma16(wrr[4 * m + 0], r0, res);
                res = mfma16(wrr[4 * m + 1], r1, res);
                res = mfma16(wrr[4 * m + 2], r2, res);
                res = mfma16(wrr[4 * m + 3], r3, res);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = (vmask >> (4 * g + r)) & 1u;
            if (qrow[r] >= 0) P[qrow[r] * PS + i16] = ok ? prelu1(fmaf(acc[r], bs, bt), ba) : 0.0f;
        }
    };

    float4 vk[MPW][NT], vr[RM][NT];
    unsigned codes[MPW];  // window codes of channels 4g..4g+3 of the lane's centre pixel
    int ipixk[MPW];       // input pixel index of the lane's centre pixel (< 2^31: guarded by the API), -1 outside the image
    bool rvalid[RM];
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int t = (wave + 4 * k) * 16 + i16;
        const int iy = ty0 + t / TW, ix = tx0 + t % TW;
        const bool valid = (iy < a.H) && (ix < a.W);
        ipixk[k] = valid ? iy * a.W + ix : -1;
        load_frags(ximg + (long)(valid ? ipixk[k] : 0) * CI, vk[k]);
        codes[k] = *reinterpret_cast<const unsigned *>(cimg + (long)(valid ? ipixk[k] : 0) * CO + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x4 resk[MPW];  // residual conv of the centre pixel: reg r = channel 4g + r
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        if (k == MPW / 2) {  // the ring fragments are requested once half of the centre registers are free again
#pragma unroll
            for (int kr = 0; kr < RM; ++kr) {
                const int u = (wave + 4 * kr) * 16 + i16;
                const int q = u < RING ? q_ring(u) : 0;
                const int pr = ty0 - 1 + q / HW2, pc = tx0 - 1 + q % HW2;
                rvalid[kr] = (u < RING) && (pr >= 0) && (pr < a.H) && (pc >= 0) && (pc < a.W);
                load_frags(rvalid[kr] ? ximg + ((long)pr * a.W + pc) * CI : ximg, vr[kr]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int mt = wave + 4 * k;
        const unsigned vmask = (unsigned)(__ballot(ipixk[k] >= 0) & 0xFFFFull);
        int qrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) qrow[r] = q_center(mt * 16 + 4 * g + r);
        resk[k] = (f32x4){0};
        project(vk[k], vmask, qrow, true, resk[k]);
    }
#pragma unroll
    for (int k = 0; k < RM; ++k) {  // halo ring
        const int mtr = wave + 4 * k;
        if (mtr * 16 < RING) {  // wave-uniform
            const unsigned vmask = (unsigned)(__ballot(rvalid[k]) & 0xFFFFull);
            int qrow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ur = mtr * 16 + 4 * g + r;
                qrow[r] = ur < RING ? q_ring(ur) : -1;
            }
            f32x4 dummy = {0};
            project(vr[k], vmask, qrow, false, dummy);
        }
    }

    // loop-invariant operands of phase B, requested before the barrier (no global load after it)
    float cs[4], ct[4], ca[4];  // transposed conv BN + PReLU: reg r of quarter g holds channel 4*(g&1) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * (g & 1) + r;
        cs[r] = a.cs[co]; ct[r] = a.ct[co]; ca[r] = a.ca[co];
    }
    // exp (8 -> 16) is evaluated as D[co][pixel]: lane = pixel, reg r = channel 4g + r
    const float4 s1 = *reinterpret_cast<const float4 *>(a.es + 4 * g);
    const float4 t1 = *reinterpret_cast<const float4 *>(a.et + 4 * g);
    const float4 al = *reinterpret_cast<const float4 *>(a.ra + 4 * g);
    const float we0 = a.we[(0 + g) * CO + i16], we1 = a.we[(4 + g) * CO + i16];  // A: row co = i16
    float wsr[6 * 4];  // stacked transposed-conv kernel as A operand, all 6 slots
#pragma unroll
    for (int slot = 0; slot < 6; ++slot)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) wsr[slot * 4 + s_] = a.ws[(slot * PF + 4 * s_ + g) * 16 + i16];
    __syncthreads();

    // ---- phase B: transposed conv 16 -> 8 (accA rows [ee|eo], accB rows [oe|oo]) -> exp -> gated residual ---
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int t = (wave + 4 * k) * 16 + i16;
        const int r_ = t / TW, c_ = t - r_ * TW;
        const int iy = ty0 + r_, ix = tx0 + c_;
        const bool ok = ipixk[k] >= 0;
        f32x4 accA = {0}, accB = {0};
#pragma unroll
        for (int slot = 0; slot < 6; ++slot) {
            const int dr = slot < 4 ? 1 - (slot >> 1) : 1, dc = 1 - (slot & 1);
            const float *pq = P + ((r_ + dr) * HW2 + (c_ + dc)) * PS + g;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                if (slot < 4) accA = mfma16(wsr[slot * 4 + s_], pq[4 * s_], accA);
                else          accB = mfma16(wsr[slot * 4 + s_], pq[4 * s_], accB);
            }
        }
        float qa[4], qb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qa[r] = prelu1(fmaf(accA[r], cs[r], ct[r]), ca[r]);
            qb[r] = prelu1(fmaf(accB[r], cs[r], ct[r]), ca[r]);
        }
        // after the transpose: regs 0,1 = first class (ci = g, 4+g), regs 2,3 = second class
        transpose4(qa[0], qa[1], qa[2], qa[3]);
        transpose4(qb[0], qb[1], qb[2], qb[3]);

        const f32x4 res = resk[k];
        const unsigned cd = codes[k];
        float *yp = yimg + (ok ? ((long)(2 * iy) * (2 * a.W) + 2 * ix) * CO : 0) + 4 * g;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {  // ee, eo, oe, oo == window code dy*2+dx
            const float b0 = cls < 2 ? qa[(cls & 1) * 2] : qb[(cls & 1) * 2];
            const float b1 = cls < 2 ? qa[(cls & 1) * 2 + 1] : qb[(cls & 1) * 2 + 1];
            f32x4 e = {0};
            e = mfma16(we0, b0, e);
            e = mfma16(we1, b1, e);
            float4 o;
            o.x = prelu1(fmaf(e[0], s1.x, t1.x) + (((cd >> 0) & 0xFFu) == (unsigned)cls ? res[0] : 0.0f), al.x);
            o.y = prelu1(fmaf(e[1], s1.y, t1.y) + (((cd >> 8) & 0xFFu) == (unsigned)cls ? res[1] : 0.0f), al.y);
            o.z = prelu1(fmaf(e[2], s1.z, t1.z) + (((cd >> 16) & 0xFFu) == (unsigned)cls ? res[2] : 0.0f), al.z);
            o.w = prelu1(fmaf(e[3], s1.w, t1.w) + (((cd >> 24) & 0xFFu) == (unsigned)cls ? res[3] : 0.0f), al.w);
            if (ok) *reinterpret_cast<float4 *>(yp + ((cls >> 1) * (2 * a.W) + (cls & 1)) * CO) = o;
        }
    }
    (void)CF;
}

// =================================================================================================
// launchers
// =================================================================================================
bool bottleneck_mfma16_supported(int Cin, int f) { return (Cin == 64 && f == 16) || (Cin == 16 && f == 4); }

hipError_t launch_bottleneck_mfma16(const BnkArgs &a0, int Cin, hipStream_t s)
{
    BnkArgs a = a0;
    a.TH = 8;
    const int Hp = (a.H + a.dil - 1) / a.dil, Wp = (a.W + a.dil - 1) / a.dil;
    const bool wide = Wp > 16;
    const int TW = wide ? 32 : 16;
    a.tiles_y = (Hp + a.TH - 1) / a.TH;
    a.tiles_x = (Wp + TW - 1) / TW;
    const long grid = (long)a.N * a.dil * a.dil * a.tiles_y * a.tiles_x;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    a.trace = (g_trace_buf && grid * 4 * 16 * 8 <= g_trace_bytes) ? g_trace_buf : nullptr;
    const double pix = (double)a.N * a.H * a.W, f = Cin / 4.0;
    ProfScope prof(Cin == 64 ? (wide ? "k_bottleneck16<32,64,16>" : "k_bottleneck16<16,64,16>")
                             : (wide ? "k_bottleneck16<32,16,4>" : "k_bottleneck16<16,16,4>"),  // = the kernel symbols
                   2.0 * pix * (Cin * f + 9.0 * f * f + f * Cin), 4.0 * 2.0 * pix * Cin, s);
    dim3 G((unsigned)grid), B(256);
    if (Cin == 64) {
        if (wide) hipLaunchKernelGGL((k_bottleneck16<32, 64, 16>), G, B, 0, s, a);
        else      hipLaunchKernelGGL((k_bottleneck16<16, 64, 16>), G, B, 0, s, a);
    } else if (Cin == 16) {
        if (wide) hipLaunchKernelGGL((k_bottleneck16<32, 16, 4>), G, B, 0, s, a);
        else      hipLaunchKernelGGL((k_bottleneck16<16, 16, 4>), G, B, 0, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool downsample_mfma16_supported(int Cin, int Cout) { return Cin == 16 && Cout == 64; }

hipError_t launch_downsample_mfma16(const DownArgs &a0, hipStream_t s)
{
    DownArgs a = a0;
    if (a.H % 2 || a.W % 2) return hipErrorInvalidValue;
    a.TH = 8;
    const int Ho = a.H / 2, Wo = a.W / 2;
    const bool wide = Wo > 16;
    const int TW = wide ? 32 : 16;
    a.tiles_y = (Ho + a.TH - 1) / a.TH;
    a.tiles_x = (Wo + TW - 1) / TW;
    const long grid = (long)a.N * a.tiles_y * a.tiles_x;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    const double opix = (double)a.N * Ho * Wo;
    ProfScope prof("k_downsample16", 2.0 * opix * (4.0 * 16 * 8 + 9.0 * 8 * 8 + 8.0 * 64),
                   4.0 * (4.0 * opix * 16 + opix * 64) + opix * 16, s);
    if (wide) hipLaunchKernelGGL(k_downsample16<32>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else      hipLaunchKernelGGL(k_downsample16<16>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

bool initial_down16_supported(int c_in) { return c_in == 1 || c_in == 3 || c_in == 4; }

// Initial + Bottleneck1_0 (H, W = IMAGE dims, divisible by 4)
hipError_t launch_initial_down16(const void *img, bool img_is_u8, int N, int H, int W, int c_in, const float *iw,
                                 const float *iscale, const float *ishift, const float *ialpha, float *y, uint8_t *code,
                                 const float *wp, const float *ps, const float *pt, const float *pa, const float *wc,
                                 const float *cs, const float *ct, const float *ca, const float *we, const float *es,
                                 const float *et, const float *ra, hipStream_t s)
{
    if (H % 4 || W % 4 || !initial_down16_supported(c_in)) return hipErrorInvalidValue;
    InitDownArgs A;
    A.img = img; A.iw = iw; A.iscale = iscale; A.ishift = ishift; A.ialpha = ialpha;
    A.d.x = nullptr; A.d.y = y; A.d.code = code;
    A.d.wp = wp; A.d.ps = ps; A.d.pt = pt; A.d.pa = pa;
    A.d.wc = wc; A.d.cs = cs; A.d.ct = ct; A.d.ca = ca;
    A.d.we = we; A.d.es = es; A.d.et = et; A.d.ra = ra;
    A.d.N = N; A.d.H = H / 2; A.d.W = W / 2;
    A.d.TH = 8;
    const int Ho = H / 4, Wo = W / 4;
    A.d.tiles_y = (Ho + 7) / 8;
    A.d.tiles_x = (Wo + 15) / 16;
    A.d.trace = nullptr;
    const long grid = (long)N * A.d.tiles_y * A.d.tiles_x;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    const double ipix = (double)N * (H / 2) * (W / 2), opix = (double)N * Ho * Wo;
    ProfScope prof("k_initial_down16",
                   2.0 * ipix * 9 * c_in * (16 - c_in) + 2.0 * opix * (4.0 * 16 * 8 + 9.0 * 8 * 8 + 8.0 * 64),
                   (img_is_u8 ? 1.0 : 4.0) * N * (double)H * W * c_in + 4.0 * opix * 64 + opix * 16, s);
    dim3 G((unsigned)grid), B(256);
#define SSAL_ID16(CINV)                                                                                      \
    if (img_is_u8) hipLaunchKernelGGL((k_initial_down16<CINV, uint8_t>), G, B, 0, s, A);                      \
    else           hipLaunchKernelGGL((k_initial_down16<CINV, float>), G, B, 0, s, A)
    if (c_in == 3) { SSAL_ID16(3); }
    else if (c_in == 4) { SSAL_ID16(4); }
    else { SSAL_ID16(1); }
#undef SSAL_ID16
    return hipGetLastError();
}

bool upsample_mfma16_supported(int Cin, int Cout) { return Cin == 64 && Cout == 16; }

hipError_t launch_upsample_mfma16(const UpArgs &a0, hipStream_t s)
{
    UpArgs a = a0;
    a.dil = 1;
    a.TH = 8;
    const bool wide = a.W > 16;
    const int TW = wide ? 32 : 16;
    a.tiles_y = (a.H + a.TH - 1) / a.TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    const long grid = (long)a.N * a.tiles_y * a.tiles_x;
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    const double pix = (double)a.N * a.H * a.W;
    ProfScope prof("k_upsample16", 2.0 * pix * (64.0 * 16 + 9.0 * 16 * 8 + 4.0 * 8 * 16 + 64.0 * 16),
                   4.0 * (pix * 64 + 4.0 * pix * 16) + pix * 16, s);
    if (wide) hipLaunchKernelGGL(k_upsample16<32>, dim3((unsigned)grid), dim3(256), 0, s, a);
    else      hipLaunchKernelGGL(k_upsample16<16>, dim3((unsigned)grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ssal
