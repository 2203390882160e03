// ssal_icnet_kernels.hip -- HIP kernels of the ICNet row (ICNET_SPEC.md), gfx950 (MI355X / CDNA4) only.
//
//   k_igemm<NT>      every convolution with Cin % 32 == 0: implicit GEMM on the fp32 matrix cores
//                    (v_mfma_f32_32x32x2_f32), fused folded batch-norm + shortcut add + ReLU epilogue,
//                    optional on-the-fly 2x bilinear up-sampling of its input (cascade feature fusion)
//   k_conv_first     3x3 / stride-2 first convolution of a branch on the 1/3/4-channel image (VALU, HBM-bound)
//   k_maxpool3x3_s2, k_ppm_pool, k_ppm_sum
//   k_upscore<K>     conv6_interp (4x bilinear) + softmax + entropy / margin / confidence + fp64 block partials:
//                    the full-resolution logits never reach HBM
//
// Bit-exactness contract (same as the ENet kernels): every conv output element is ONE fp32 fmaf chain over
// (kh, kw, ci) ascending -- the fp32 MFMA is exactly such a chain over its k index -- so results equal the parity
// oracle bit for bit.  Out-of-image taps contribute fmaf(0, w, acc) == acc.
#include "ssal_icnet.h"
#include "ssal_internal.h"
#include "ssal_mfma.h"
#include "ssal_prof.h"
#include "ssal_score.h"

#include <float.h>
#include <string.h>

namespace ssal {

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 matrix cores.
//   D[pixel][cout] = sum_k A[pixel][k] * B[k][cout],  k = (kh*KW + kw)*Cin + ci  ascending.
// Workgroup = 4 waves, tile = 128 pixels x 32*NT couts; wave w owns pixel rows [32w, 32w+32) and all 32*NT columns
// (NT accumulator tiles of 32x32).  K is walked in chunks of 32 input channels of one tap: the A chunk is 128 pixels
// x 128 B (one full line per pixel, coalesced), the B chunk is 32*NT rows x 128 B, contiguous in the re-laid-out
// kernel [tap][Cin/32][CoutP][32].  Both go global -> registers -> LDS (two buffers: the loads of chunk t+1 are in
// flight while chunk t is multiplied), rows padded to 36 floats so that the ds_read_b128 fragment reads
// (lane = row, one quad per lane half) are bank-conflict free.  The k order inside a row is permuted (igemm_kpos) so
// that the quad a lane half reads is already the MFMA operand sequence in ascending k order.
// ------------------------------------------------------------------------------------------------
constexpr int IG_BM = 128, IG_LDK = 36;
// Position of channel k (0..31) inside a 32-channel chunk row, in LDS and in the re-laid-out kernel: every group of 8
// is stored as [k0 k2 k4 k6 | k1 k3 k5 k7], so that ONE ds_read_b128 per lane half (h = 0: first quad, h = 1: second)
// delivers, register by register, exactly the (k = 2s | k = 2s+1) lane-half pairs the MFMA steps s = 0..3 consume in
// ascending k order -- no register re-pairing (v_permlane32_swap) between the read and the MFMA.
__host__ __device__ constexpr int igemm_kpos(int k) { return (k & ~7) | ((k & 1) << 2) | ((k & 7) >> 1); }
constexpr unsigned IG_OOB = 0xFFFFFFFFu;  // a byte offset no tensor reaches: raw buffer loads return 0, stores are dropped

// Addressing: every tensor is reached through a raw buffer resource (base in SGPRs, range = tensor bytes) plus a
// 32-bit per-lane byte offset.  A pixel row's offset is computed ONCE; per K-chunk it only receives a wave-uniform
// increment (tap displacement + channel chunk), and taps that fall into the SAME zero padding (or rows past the end of
// the tensor) use the out-of-range offset, which the hardware answers with zeros -- no branches, no 64-bit math.
// UP2: 0 = plain; 1 = the conv runs on resize_bilinear(x, 2x), four neighbours fetched per A row (any shape); 2 = the same
// for stride 1, even dilation / padding and Wo % 4 == 0 (ICNet's conv_sub4 / conv_sub2): a thread owns four ADJACENT output
// pixels of one image row, whose taps interpolate from 3 source columns x 2 source rows -- 6 loads per chunk instead of 16,
// same four source values and the same lerp formula per pixel, i.e. the same bits.
// DUAL: two convolutions into one output (a bottleneck's projection shortcut + its 1x1 "increase"; score path only): pass 0
// runs the SECOND source (a0.x2: 1x1, stride a0.stride2, its own folded batch-norm) through the same K loop and parks
// y_p = fmaf(acc, scale2, shift2) in registers, pass 1 runs the main convolution and adds y_p where the separate launches
// add the shortcut tensor they read back -- the same arithmetic, without the shortcut's write + read.
// SB: ONE LDS buffer instead of two (two barriers per chunk: after the last fragment read, after the write of the next
// chunk): 36.9 KB at NT = 4 and <= 168 VGPRs -- three workgroups per CU instead of two.  Measured (profiles/
// r05_ab_igemm_single_lds_buffer.txt): the up-sampling form, whose load + interpolation phases are the longest, gains 3 %
// (conv_sub2 / conv_sub4); the plain form loses 7 % at NT = 4 and gains 1 % at NT = 2; four workgroups per CU (128 VGPRs:
// 20-37 spilled) lose 4 %.  Used for UP2 = 2 and, since ICNet runs on one chain, for the plain form at NT = 1 / 2 (knob ig_sb).
template <int NT, int UP2, bool DUAL = false, bool SB = false>
__global__ __launch_bounds__(256, SB ? 3 : 1) void k_igemm(IgemmArgs a0)
{
    IgemmArgs a = a0;
    constexpr int BM = IG_BM, BN = 32 * NT, LDK = IG_LDK;
    // The kernel tiles (B) are always stored permuted (free: done once at commit).  The activation tile (A) is
    // permuted while it is written to LDS when that pays: two 8-byte LDS writes per quad instead of one 16-byte
    // write, against 2 register swaps per fragment read -- a win when the fragment feeds several column tiles.
    constexpr bool A_PERMUTED = NT >= 2;
#ifdef SSAL_MEASURE
    constexpr bool MIDWRITE = false;  // the phase trace / phase ablation times the write phase on its own
#else
    constexpr bool MIDWRITE = NT >= 2 && !SB;  // NT = 1: 16 MFMAs per chunk are over before the loads are back
#endif
    constexpr int WG_A = NT >= 4 ? 1 : 2, WG_B = WG_A + 1;  // 8-k groups after which the A rows / the kernel rows are written
    constexpr int NBUF = SB ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float smem[NBUF * (BM + BN) * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    int tile = blockIdx.x;
    if (a.xcd_chunk) {
        tile = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
        if (tile >= a.ntiles) return;  // whole workgroup, before any barrier
    }
    const int tn = tile % a.tiles_n, tm = tile / a.tiles_n;
    const int m0 = tm * BM;  // M < 2^31 (launcher): 32-bit pixel indices keep the row decode to 32-bit divisions
    const int n0 = tn * BN;
    const int M = (int)a.M;

    f32x16 acc[NT];
    float yp[DUAL ? NT : 1][16];  // DUAL: the shortcut branch's output for this lane's 16 pixels x NT columns
#ifdef SSAL_PHASE_TRACE  // tools/igemm_trace.py: per-wave cycle totals of the phases of the K loop
    unsigned long long tr_t0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr_gl = 0, tr_mm = 0, tr_vm = 0, tr_wr = 0, tr_bar = 0, tr_a, tr_b;
#define TR_MARK(acc_)  do { tr_b = __builtin_amdgcn_s_memtime(); acc_ += tr_b - tr_a; tr_a = tr_b; } while (0)
#else
#define TR_MARK(acc_)  do { } while (0)
#endif
#ifdef SSAL_PHASE_TRACE
    unsigned long long tr_loop = 0, tr_loop_end = 0;
#endif
#pragma unroll 1
    for (int pass = DUAL ? 0 : 1; pass < 2; ++pass) {
    if (DUAL) {
        a = a0;
        if (pass == 0) {  // the shortcut branch: 1x1, no padding, its own stride / input
            a.x = a0.x2; a.wt = a0.wt2; a.Cin = a0.Cin2; a.H = a0.H2; a.W = a0.W2; a.stride = a0.stride2;
            a.KH = 1; a.KW = 1; a.dil = 1; a.pad_t = 0; a.pad_l = 0;
        }
    }
    const int Hs = UP2 ? a.H >> 1 : a.H, Ws = UP2 ? a.W >> 1 : a.W;  // dims of the tensor in memory (UP2: int, 0 / 1 / 2)
    const rsrc_t xrs = make_rsrc(a.x, (unsigned)((long)a.N * Hs * Ws * a.Cin * 4));
    const rsrc_t wrs = make_rsrc(a.wt, (unsigned)((long)a.KH * a.KW * a.Cin * a.CoutP * 4));

    // ---- per-thread A rows: (tid >> 3) + 32 i (UP2 == 2: 4 (tid >> 3) + i), channel quad tid & 7 ----
    const int col4 = tid & 7;
    auto arow = [&](int i) { return UP2 == 2 ? 4 * (tid >> 3) + i : (tid >> 3) + 32 * i; };
    int iyb[4], ixb[4];
    unsigned nbase[4];  // byte offset of the row's image
    bool mv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + arow(i);
        mv[i] = m < M;
        const unsigned mm = mv[i] ? (unsigned)m : 0u;
        const unsigned t = mm / (unsigned)a.Wo;
        const int ox = (int)(mm - t * (unsigned)a.Wo);
        const unsigned n = t / (unsigned)a.Ho;
        const int oy = (int)(t - n * (unsigned)a.Ho);
        iyb[i] = oy * a.stride - a.pad_t;
        ixb[i] = ox * a.stride - a.pad_l;
        nbase[i] = (unsigned)n * (unsigned)(Hs * Ws * a.Cin * 4) + 16u * col4;
    }
    const int cpt = a.Cin >> 5;             // chunks per tap
    const int nchunks = a.KH * a.KW * cpt;
    const unsigned rowb = (unsigned)(Ws * a.Cin * 4), pixb = (unsigned)(a.Cin * 4);

    float4 rb[NT];
    // Loader state: the NEXT chunk to request is (tap = (ld_kh, ld_kw), channel chunk ld_cc).  The per-row byte offsets
    // (and SAME-padding validity) depend on the tap only: they are computed when a new tap starts and kept in registers;
    // within a tap a chunk only adds the wave-uniform 128 * ld_cc, which rides in the instruction's scalar offset
    // (excluded from the range check, so an out-of-range row offset stays out of range).  A 1x1 convolution computes its
    // offsets once; a 3x3 one on 256 channels once per 8 chunks.
    constexpr int NOFF = UP2 == 1 ? 16 : UP2 == 2 ? 6 : 4;
    unsigned aoff[NOFF];
    float4 rq[NOFF];  // the loaded quads of the next chunk (UP2 = 1: the four neighbours tl, tr, bl, br of each row;
                      // UP2 = 2: source rows y0, y1 x source columns A, B, C shared by the thread's four pixels)
    float lyv[4], lxv[4];  // UP2: the two interpolation weights of each row (0 or 0.5)
    bool pok[2] = {false, false};  // UP2 = 2: pixel pairs (0, 1) / (2, 3) inside the image for the current tap
    int ld_cc = 0, ld_kh = 0, ld_kw = 0;
    unsigned ld_wofs = (unsigned)n0 * 128u;  // wave-uniform byte offset of the kernel rows of the next chunk
    auto tap_offsets = [&]() {
        if (UP2 == 2) {
            // the four pixels sit in one image row at ix0 .. ix0 + 3 with ix0 even (launcher): pairs (ix0, ix0 + 1) and
            // (ix0 + 2, ix0 + 3) read source columns (A, B) and (B, C), A = ix0 >> 1, B / C its right neighbours clamped
            // as the per-pixel rule x1 = min(x0 + 1, Ws - 1) clamps them; a pair is inside the image or outside as a whole
            const int iy = iyb[0] + ld_kh * a.dil, ix0 = ixb[0] + ld_kw * a.dil;
            const bool rowok = mv[0] && iy >= 0 && iy < a.H;
            pok[0] = rowok && ix0 >= 0 && ix0 < a.W;
            pok[1] = rowok && ix0 + 2 >= 0 && ix0 + 2 < a.W;
            const int y0 = iy >> 1, y1 = min(y0 + 1, Hs - 1);
            const int xa = ix0 >> 1, xb = min(xa + 1, Ws - 1), xc = min(xa + 2, Ws - 1);
            lyv[0] = (iy & 1) ? 0.5f : 0.0f;
            const unsigned b0 = nbase[0] + (unsigned)y0 * rowb, b1 = nbase[0] + (unsigned)y1 * rowb;
            aoff[0] = pok[0] ? b0 + (unsigned)xa * pixb : IG_OOB;
            aoff[1] = (pok[0] || pok[1]) ? b0 + (unsigned)xb * pixb : IG_OOB;
            aoff[2] = pok[1] ? b0 + (unsigned)xc * pixb : IG_OOB;
            aoff[3] = pok[0] ? b1 + (unsigned)xa * pixb : IG_OOB;
            aoff[4] = (pok[0] || pok[1]) ? b1 + (unsigned)xb * pixb : IG_OOB;
            aoff[5] = pok[1] ? b1 + (unsigned)xc * pixb : IG_OOB;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = iyb[i] + ld_kh * a.dil, ix = ixb[i] + ld_kw * a.dil;
            const bool ok = mv[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            if (UP2 == 0) {
                aoff[i] = ok ? nbase[i] + (unsigned)iy * rowb + (unsigned)ix * pixb : IG_OOB;
            } else if (UP2 == 1) {
                // tf.image.resize_bilinear(src, 2x), legacy mapping src = dst * 0.5 (ICNET_SPEC "bilinear resize");
                // out-of-image taps read four zeros and interpolate to an exact zero
                const int y0 = iy >> 1, x0 = ix >> 1;
                const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
                lyv[i] = (iy & 1) ? 0.5f : 0.0f;
                lxv[i] = (ix & 1) ? 0.5f : 0.0f;
                const unsigned b0 = nbase[i];
                aoff[4 * i + 0] = ok ? b0 + (unsigned)y0 * rowb + (unsigned)x0 * pixb : IG_OOB;
                aoff[4 * i + 1] = ok ? b0 + (unsigned)y0 * rowb + (unsigned)x1 * pixb : IG_OOB;
                aoff[4 * i + 2] = ok ? b0 + (unsigned)y1 * rowb + (unsigned)x0 * pixb : IG_OOB;
                aoff[4 * i + 3] = ok ? b0 + (unsigned)y1 * rowb + (unsigned)x1 * pixb : IG_OOB;
            }
        }
    };
    // live = false (wave-uniform; the call past the last chunk): every offset is out of range, the loads return zeros and
    // nothing reads them.  Keeping the call unconditional keeps the loaded registers out of a loop-carried PHI: with
    // `if (more) gload()` the compiler copies them right behind the loads -- i.e. waits for them BEFORE the MFMAs of the
    // current chunk instead of after.
    auto gload = [&](bool live) {
        if (live && ld_cc == 0) tap_offsets();  // wave-uniform
        const unsigned cofs = 128u * (unsigned)ld_cc;
#pragma unroll
        for (int q = 0; q < NOFF; ++q) rq[q] = bload4(xrs, live ? aoff[q] : IG_OOB, cofs);
#pragma unroll
        for (int j = 0; j < NT; ++j) rb[j] = bload4(wrs, live ? 16u * (unsigned)(tid + 256 * j) : IG_OOB, ld_wofs);
        ld_wofs += (unsigned)a.CoutP * 128u;
        if (++ld_cc == cpt) {
            ld_cc = 0;
            if (++ld_kw == a.KW) { ld_kw = 0; ++ld_kh; }
        }
    };
    auto lds_write_a = [&](int buf) {
        float *As = smem + buf * (BM + BN) * LDK;
        float4 ra[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (UP2 == 0) {
                ra[i] = rq[i];
            } else {  // the interpolation happens here, i.e. after the loads have had the matrix section to land
                const int pr_ = i >> 1;  // UP2 == 2: pixel pair -> source columns (A, B) or (B, C)
                const float4 tl = UP2 == 2 ? rq[pr_] : rq[4 * i], tr = UP2 == 2 ? rq[pr_ + 1] : rq[4 * i + 1];
                const float4 bl = UP2 == 2 ? rq[3 + pr_] : rq[4 * i + 2], br = UP2 == 2 ? rq[4 + pr_] : rq[4 * i + 3];
                const float lx = UP2 == 2 ? ((i & 1) ? 0.5f : 0.0f) : lxv[i], ly = UP2 == 2 ? lyv[0] : lyv[i];
                auto lerp2 = [&](float ctl, float ctr, float cbl, float cbr) {
                    const float top = ctl + (ctr - ctl) * lx;
                    const float bot = cbl + (cbr - cbl) * lx;
                    return top + (bot - top) * ly;
                };
                ra[i] = make_float4(lerp2(tl.x, tr.x, bl.x, br.x), lerp2(tl.y, tr.y, bl.y, br.y),
                                    lerp2(tl.z, tr.z, bl.z, br.z), lerp2(tl.w, tr.w, bl.w, br.w));
                // a pair outside the image is an exact zero (the per-pixel form reads four zeros there); column B is shared
                if (UP2 == 2 && !pok[pr_]) ra[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (A_PERMUTED) {
                // channels 4*col4 .. +3 = (k, k+1, k+2, k+3) with k = 0 or 4 (mod 8): k and k+2 are neighbours in
                // the permuted row, so are k+1 and k+3
                float *ap = As + arow(i) * LDK + 8 * (col4 >> 1) + 2 * (col4 & 1);
                *reinterpret_cast<float2 *>(ap) = make_float2(ra[i].x, ra[i].z);
                *reinterpret_cast<float2 *>(ap + 4) = make_float2(ra[i].y, ra[i].w);
            } else {
                *reinterpret_cast<float4 *>(As + arow(i) * LDK + 4 * col4) = ra[i];
            }
        }
    };
    auto lds_write_b = [&](int buf) {
        float *Bs = smem + buf * (BM + BN) * LDK + BM * LDK;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4 *>(Bs + (idx >> 3) * LDK + 4 * (idx & 7)) = rb[j];
        }
    };
    auto lds_write = [&](int buf) { lds_write_a(buf); lds_write_b(buf); };

#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
    gload(true);
    lds_write(0);
    __syncthreads();
#ifdef SSAL_PHASE_TRACE
    tr_loop = __builtin_amdgcn_s_memtime();
    tr_a = tr_loop;
#endif
    for (int t = 0; t < nchunks; ++t) {
        const bool more = t + 1 < nchunks;
        // first fragments of this chunk: requested BEFORE the address arithmetic of the next chunk's loads, which hides
        // their LDS latency
        const float *As = smem + (SB ? 0 : t & 1) * (BM + BN) * LDK + (32 * wave + r) * LDK + 4 * h;
        const float *Bs = smem + (SB ? 0 : t & 1) * (BM + BN) * LDK + BM * LDK + r * LDK + 4 * h;
        // fragments of 8-k group g+1 are requested before the 4*NT MFMAs of group g: the LDS latency hides behind them
        float4 af[2], bf[2][NT];
        af[0] = *reinterpret_cast<const float4 *>(As);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *reinterpret_cast<const float4 *>(Bs + 32 * nt * LDK);
        __builtin_amdgcn_sched_barrier(0);
#ifdef SSAL_MEASURE
        gload(more && !(a.ablate & 2));
#else
        if (NT >= 2) gload(more);     // measured: -3 % (NT = 4), -4 % (NT = 2); NT = 1 (natural-order A rows, no re-pairing
        else if (more) gload(true);   // copies) is 4 % faster with the plain conditional form
#endif
        TR_MARK(tr_gl);
        // the loads of chunk t+1 stay in flight across the matrix section: nothing that consumes them (the register
        // re-pairing of the permuted LDS writes, the 2x interpolation) may be scheduled above this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#ifdef SSAL_MEASURE
            if (a.ablate & 1) break;  // timing only: no fragment reads, no MFMAs
#endif
            const int c = g & 1, nx = c ^ 1;
            if (g < 3) {
                af[nx] = *reinterpret_cast<const float4 *>(As + 8 * (g + 1));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bf[nx][nt] = *reinterpret_cast<const float4 *>(Bs + 32 * nt * LDK + 8 * (g + 1));
            }
            if (!A_PERMUTED) {  // natural k order in the A rows: re-pair the registers of the ONE A fragment instead
                swap32(af[c].x, af[c].y);  // .x = (k0 | k1), .y = (k4 | k5)
                swap32(af[c].z, af[c].w);  // .z = (k2 | k3), .w = (k6 | k7)
                const float t = af[c].y;
                af[c].y = af[c].z;         // -> step order x, y, z, w = (k0|k1), (k2|k3), (k4|k5), (k6|k7)
                af[c].z = t;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = mfma32(af[c].x, bf[c][nt].x, acc[nt]);
                acc[nt] = mfma32(af[c].y, bf[c][nt].y, acc[nt]);
                acc[nt] = mfma32(af[c].z, bf[c][nt].z, acc[nt]);
                acc[nt] = mfma32(af[c].w, bf[c][nt].w, acc[nt]);
            }
            // Chunk t+1 goes to the other LDS buffer (free since the barrier that ended chunk t-1) from INSIDE the matrix
            // section: its loads have had half of the section to land, the re-pairing copies / the interpolation and the
            // LDS write latency hide behind the remaining MFMAs, and nothing but the barrier is left after the last one.
            // (NT >= 2 writes unconditionally: past the last chunk the registers hold zeros and the buffer is dead.)
            if (MIDWRITE && g == WG_A) {
                __builtin_amdgcn_sched_barrier(0);
                if (NT >= 2 || more) lds_write_a((t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MIDWRITE && g == WG_B) {
                __builtin_amdgcn_sched_barrier(0);
                if (NT >= 2 || more) lds_write_b((t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        TR_MARK(tr_mm);
        __builtin_amdgcn_sched_barrier(0);
#ifdef SSAL_PHASE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TR_MARK(tr_vm);
#endif
        if (SB) __syncthreads();  // every wave has read its last fragment of chunk t
        if (!MIDWRITE) {
#ifdef SSAL_MEASURE
            if (more && !(a.ablate & 4)) lds_write(SB ? 0 : (t + 1) & 1);
#else
            if (more) lds_write(SB ? 0 : (t + 1) & 1);
#endif
        }
        TR_MARK(tr_wr);
        __syncthreads();
        TR_MARK(tr_bar);
    }
#ifdef SSAL_PHASE_TRACE
    tr_loop_end = __builtin_amdgcn_s_memtime();
#endif
    if (DUAL && pass == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float sc2 = a0.scale2[n0 + 32 * nt + r], sh2 = a0.shift2[n0 + 32 * nt + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) yp[DUAL ? nt : 0][i] = fmaf(acc[nt][i], sc2, sh2);
        }
    }
    }  // pass

    // ---- epilogue: folded batch-norm, shortcut add, ReLU, store (128-B rows per lane half) --------
    const unsigned ybytes = (unsigned)(a.M * a.Cout * 4);
    const rsrc_t yrs = make_rsrc(a.y, ybytes);
    const rsrc_t rrs = make_rsrc(a.res ? a.res : a.y, ybytes);
    unsigned moff[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * h;
        moff[i] = m < M ? (unsigned)m * (unsigned)(a.Cout * 4) : IG_OOB;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + 32 * nt + r;
        const float sc = a.scale[co], sh = a.shift[co];
        const bool cok = co < a.Cout;
        float rv[16];
        if (DUAL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = yp[DUAL ? nt : 0][i];
        } else if (a.res) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = bload(rrs, (cok && moff[i] != IG_OOB) ? moff[i] + 4u * co : IG_OOB, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = fmaf(acc[nt][i], sc, sh);
            if (DUAL || a.res) v = v + rv[i];
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                  (cok && moff[i] != IG_OOB) ? moff[i] + 4u * co : IG_OOB, 0, 0);
        }
    }
#ifdef SSAL_PHASE_TRACE
    if (a.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *p = a.trace + ((long)blockIdx.x * 4 + wave) * 16;
        p[0] = tr_t0; p[1] = tr_loop; p[2] = tr_gl; p[3] = tr_mm; p[4] = tr_vm; p[5] = tr_wr; p[6] = tr_bar;
        p[7] = tr_loop_end; p[8] = __builtin_amdgcn_s_memtime(); p[9] = (unsigned long long)nchunks;
        p[12] = tr_r0; p[13] = __builtin_amdgcn_s_memrealtime();
        p[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        p[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / dilation 1 convolution on 32 input channels (conv1_2_3x3, conv1_3_3x3, conv2_x_3x3): the 9 taps of
// neighbouring pixels read the same input pixels, so instead of re-staging a 128-pixel A chunk per tap (k_igemm) the
// workgroup stages the (8+2) x (16+2) pixel window ONCE and then runs the 144 MFMAs of its K loop (9 taps x 32
// channels) without a single global load or barrier: fragments are read from the window at the tap's offset.
// Workgroups are persistent: the 9 x 32 x 32 kernel slice of their column tile is staged once, and the window of the
// next spatial tile is requested before the K loop of the current one.  Same arithmetic order as k_igemm (taps ascending,
// channels ascending), same permuted-k rows (no operand re-pairing), same epilogue.
// Workgroup = 4 waves, tile = 8 x 16 output pixels x 32 output channels; wave w owns tile rows 2w, 2w+1.
// LDS 67 KB = two workgroups per CU.
// ------------------------------------------------------------------------------------------------
constexpr int HT_H = 8, HT_W = 16, HT_WW = HT_W + 2, HT_WP = (HT_H + 2) * HT_WW;  // 180 window pixels

struct HaloArgs {
    const float *x, *wt, *scale, *shift, *res;
    float *y;
    int N, H, W, Cout, CoutP, relu;
    int tiles_x, tiles_y, tiles_n;
};

__global__ __launch_bounds__(256) void k_conv3x3_c32(HaloArgs a)
{
    constexpr int LDK = IG_LDK;
    __shared__ __attribute__((aligned(16))) float As[HT_WP * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[9 * 32 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // persistent workgroup: keeps ONE column tile's kernel slice in LDS and walks spatial tiles
    // sp = first, first + step, ...; the window of tile i+1 is in flight while tile i is multiplied
    const int tn = blockIdx.x % a.tiles_n, n0 = tn * 32;
    const int step = gridDim.x / a.tiles_n;
    int sp = blockIdx.x / a.tiles_n;
    const int nsp = a.N * a.tiles_y * a.tiles_x;
    if (sp >= nsp) return;  // whole workgroup, before any barrier

    const unsigned img = (unsigned)(a.H * a.W * 32 * 4);
    const rsrc_t wrs = make_rsrc(a.wt, (unsigned)(9 * a.CoutP * 32 * 4));
    constexpr int AQ = HT_WP * 8, AIT = (AQ + 255) / 256;  // float4 quads of the window
    float4 sa[AIT];
    auto decode = [&](int t, int &n, int &ty0, int &tx0) {
        const int tx = t % a.tiles_x;
        const int q = t / a.tiles_x;
        tx0 = tx * HT_W;
        ty0 = (q % a.tiles_y) * HT_H;
        n = q / a.tiles_y;
    };
    // unconditional buffer loads; outside the image = out-of-range offset = zeros.  live = false (wave-uniform; the call
    // after the last tile): every offset out of range.  The call itself must stay unconditional: under `if (next tile)`
    // the loaded registers become loop-carried PHIs whose copies -- and the wait for the loads -- the compiler places
    // right behind the loads, i.e. BEFORE the K loop they are meant to overlap.
    auto load_window = [&](int t, bool live) {
        int n, ty0, tx0;
        decode(t, n, ty0, tx0);
        const rsrc_t xrs = make_rsrc(a.x + (long)n * a.H * a.W * 32, img);
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int e = tid + 256 * it;
            const int px = min(e >> 3, HT_WP - 1), q = e & 7;
            const int iy = ty0 - 1 + px / HT_WW, ix = tx0 - 1 + px % HT_WW;
            const bool ok = live && e < AQ && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            sa[it] = bload4(xrs, ok ? (unsigned)((iy * a.W + ix) * 128 + 16 * q) : IG_OOB, 0);
        }
    };
    load_window(sp, true);
    {
        float4 sb[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) sb[t] = bload4(wrs, 16u * (unsigned)tid, (unsigned)(t * a.CoutP + n0) * 128u);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            *reinterpret_cast<float4 *>(Bs + (t * 32 + (tid >> 3)) * LDK + 4 * (tid & 7)) = sb[t];
    }
    const float bsc = a.scale[n0 + r], bsh = a.shift[n0 + r];
    const int co = n0 + r;
    const bool cok = co < a.Cout;
    const unsigned ybytes = (unsigned)(a.H * a.W * a.Cout * 4);
    const int pr = 2 * wave + (r >> 4), pc = r & 15;
    const float *Ab = As + (pr * HT_WW + pc) * LDK + 4 * h;
    const float *Bb = Bs + r * LDK + 4 * h;

    for (; sp < nsp; sp += step) {
        int n, ty0, tx0;
        decode(sp, n, ty0, tx0);
        __syncthreads();  // every wave has finished reading the previous window
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int e = tid + 256 * it;
            if (e < AQ) {  // permuted-k row (igemm_kpos): (k, k+2) and (k+1, k+3) are neighbours
                float *ap = As + (e >> 3) * LDK + 8 * ((e & 7) >> 1) + 2 * (e & 1);
                *reinterpret_cast<float2 *>(ap) = make_float2(sa[it].x, sa[it].z);
                *reinterpret_cast<float2 *>(ap + 4) = make_float2(sa[it].y, sa[it].w);
            }
        }
        __syncthreads();
        {
            const bool more = sp + step < nsp;
            load_window(more ? sp + step : sp, more);  // in flight during the K loop below
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- K loop: 9 taps x 4 groups of 8 channels, fragments one group ahead, no global load, no barrier ----
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        float4 af[2], bf[2];
        af[0] = *reinterpret_cast<const float4 *>(Ab);
        bf[0] = *reinterpret_cast<const float4 *>(Bb);
#pragma unroll
        for (int s = 0; s < 36; ++s) {  // s = tap * 4 + group
            const int c = s & 1, nx = c ^ 1;
            if (s + 1 < 36) {
                const int t1 = (s + 1) >> 2, g1 = (s + 1) & 3;
                af[nx] = *reinterpret_cast<const float4 *>(Ab + ((t1 / 3) * HT_WW + (t1 % 3)) * LDK + 8 * g1);
                bf[nx] = *reinterpret_cast<const float4 *>(Bb + t1 * 32 * LDK + 8 * g1);
            }
            acc = mfma32(af[c].x, bf[c].x, acc);
            acc = mfma32(af[c].y, bf[c].y, acc);
            acc = mfma32(af[c].z, bf[c].z, acc);
            acc = mfma32(af[c].w, bf[c].w, acc);
        }

        // ---- epilogue ------------------------------------------------------------------------------------
        const rsrc_t yrs = make_rsrc(a.y + (long)n * a.H * a.W * a.Cout, ybytes);
        const rsrc_t rrs = make_rsrc((a.res ? a.res : a.y) + (long)n * a.H * a.W * a.Cout, ybytes);
        unsigned moff[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int oy = ty0 + 2 * wave + (m >> 4), ox = tx0 + (m & 15);
            moff[i] = (cok && oy < a.H && ox < a.W) ? (unsigned)(((oy * a.W + ox) * a.Cout + co) * 4) : IG_OOB;
        }
        float rv[16];
        if (a.res) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = bload(rrs, moff[i], 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = fmaf(acc[i], bsc, bsh);
            if (a.res) v = v + rv[i];
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, moff[i], 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution on the 2x bilinearly up-sampled input, 128 channels -> at most 32 outputs (ICNET_SPEC
// sub12_sum_interp -> conv6_cls): the generic UP2 loader fetches four source pixels and interpolates them for every
// chunk of every output pixel -- 16 loads + ~150 VALU per chunk against 16 MFMAs.  Here a workgroup owns 8 x 16 output
// pixels, stages the 5 x 9 SOURCE pixels they interpolate from ONCE in LDS (permuted-k rows, so the interpolated quad
// of a lane half is already the MFMA operand sequence; the lerp is element-wise and commutes with the permutation) and
// builds its A fragments from LDS: 4 ds_read_b128 + 12 lerps per 4 MFMAs.  Same arithmetic as k_igemm<1, true>.
// ------------------------------------------------------------------------------------------------
constexpr int CU_SH = HT_H / 2 + 1, CU_SW = HT_W / 2 + 1, CU_SP = CU_SH * CU_SW, CU_LDK = 128 + 4;  // 5 x 9 source pixels

struct ClsUpArgs {
    const float *x, *wt, *scale, *shift;
    float *y;
    int N, Hs, Ws, Cout, relu;  // Hs, Ws: source dims; the output is [N, 2Hs, 2Ws, Cout]
    int tiles_x, tiles_y;
};

// Persistent (round 5): 768 workgroups walk the tiles; the kernel is staged ONCE per workgroup and the source window of the
// next tile is requested into registers before the current tile's MFMAs (k_conv3x3_c32's scheme), so that a tile no longer
// starts with an exposed HBM round trip.
__global__ __launch_bounds__(256) void k_conv1x1_up2_c128(ClsUpArgs a)
{
    __shared__ __attribute__((aligned(16))) float Ss[CU_SP * CU_LDK];
    __shared__ __attribute__((aligned(16))) float Bs[4 * 32 * IG_LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int H = 2 * a.Hs, W = 2 * a.Ws;
    constexpr int SQ = CU_SP * 32, SIT = (SQ + 255) / 256;  // float4 quads of the window (32 per pixel)
    const rsrc_t wrs = make_rsrc(a.wt, 4u * 32u * 128u);
    float4 ss[SIT];
    int n = 0, oy0 = 0, ox0 = 0;
    auto decode = [&](int t, int &n_, int &oy_, int &ox_) {
        const int tx = t % a.tiles_x; t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        n_ = t / a.tiles_y; oy_ = ty * HT_H; ox_ = tx * HT_W;  // even
    };
    auto request = [&](int t) {  // the 5 x 9 source pixels of tile t -> registers (out of range past the last tile: zeros nobody reads)
        int n_, oy_, ox_;
        decode(t < ntiles ? t : 0, n_, oy_, ox_);
        const rsrc_t xrs = make_rsrc(a.x + (long)n_ * a.Hs * a.Ws * 128, (unsigned)(a.Hs * a.Ws * 512));
        const int sy0 = oy_ >> 1, sx0 = ox_ >> 1;
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + 256 * it;
            const int px = min(e >> 5, CU_SP - 1), q = e & 31;
            const int sy = min(sy0 + px / CU_SW, a.Hs - 1), sx = min(sx0 + px % CU_SW, a.Ws - 1);  // clamped = y1 / x1 rule
            ss[it] = bload4(xrs, (e < SQ && t < ntiles) ? (unsigned)((sy * a.Ws + sx) * 512 + 16 * q) : IG_OOB, 0);
        }
    };
    auto park = [&]() {  // registers -> LDS, permuted-k rows (igemm_kpos)
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + 256 * it;
            if (e < SQ) {
                float *sp = Ss + (e >> 5) * CU_LDK + 8 * ((e & 31) >> 1) + 2 * (e & 1);
                *reinterpret_cast<float2 *>(sp) = make_float2(ss[it].x, ss[it].z);
                *reinterpret_cast<float2 *>(sp + 4) = make_float2(ss[it].y, ss[it].w);
            }
        }
    };
    int t = blockIdx.x;
    if (t >= ntiles) return;  // whole workgroup, before any barrier
    request(t);
    {
        float4 sb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sb[q] = bload4(wrs, 16u * (unsigned)tid, (unsigned)q * 4096u);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(Bs + (q * 32 + (tid >> 3)) * IG_LDK + 4 * (tid & 7)) = sb[q];
    }
    const float bsc = a.scale[r], bsh = a.shift[r];
    park();
    __syncthreads();
    // ---- this lane's output pixel and its four source pixels (tf.image.resize_bilinear, legacy mapping) ----
    const int pr = 2 * wave + (r >> 4), pc = r & 15;
    const float ly = (pr & 1) ? 0.5f : 0.0f, lx = (pc & 1) ? 0.5f : 0.0f;
    const int y0 = pr >> 1, x0 = pc >> 1;  // window coordinates; the +1 neighbours were clamped while staging
    const float *ptl = Ss + (y0 * CU_SW + x0) * CU_LDK + 4 * h;
    const float *ptr_ = ptl + CU_LDK, *pbl = ptl + CU_SW * CU_LDK, *pbr = pbl + CU_LDK;
    const float *Bb = Bs + r * IG_LDK + 4 * h;
    auto lerp4 = [&](const float4 &tl, const float4 &tr, const float4 &bl, const float4 &br) {
        auto l1 = [&](float ctl, float ctr, float cbl, float cbr) {
            const float top = ctl + (ctr - ctl) * lx;
            const float bot = cbl + (cbr - cbl) * lx;
            return top + (bot - top) * ly;
        };
        return make_float4(l1(tl.x, tr.x, bl.x, br.x), l1(tl.y, tr.y, bl.y, br.y), l1(tl.z, tr.z, bl.z, br.z),
                           l1(tl.w, tr.w, bl.w, br.w));
    };
    const bool cok = r < a.Cout;
    for (; t < ntiles; t += gridDim.x) {
        decode(t, n, oy0, ox0);
        request(t + (int)gridDim.x);  // the next tile's window travels under this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // 8-channel groups in ascending order: chunk g / 4, group g % 4
            const float4 tl = *reinterpret_cast<const float4 *>(ptl + 8 * g), tr = *reinterpret_cast<const float4 *>(ptr_ + 8 * g);
            const float4 bl = *reinterpret_cast<const float4 *>(pbl + 8 * g), br = *reinterpret_cast<const float4 *>(pbr + 8 * g);
            const float4 bf = *reinterpret_cast<const float4 *>(Bb + (g >> 2) * 32 * IG_LDK + 8 * (g & 3));
            const float4 af = lerp4(tl, tr, bl, br);
            acc = mfma32(af.x, bf.x, acc);
            acc = mfma32(af.y, bf.y, acc);
            acc = mfma32(af.z, bf.z, acc);
            acc = mfma32(af.w, bf.w, acc);
        }
        // ---- epilogue ----
        const rsrc_t yrs = make_rsrc(a.y + (long)n * H * W * a.Cout, (unsigned)(H * W * a.Cout * 4));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int oy = oy0 + 2 * wave + (m >> 4), ox = ox0 + (m & 15);
            float v = fmaf(acc[i], bsc, bsh);
            if (a.relu) v = v > 0.0f ? v : 0.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                  (cok && oy < H && ox < W) ? (unsigned)(((oy * W + ox) * a.Cout + r) * 4) : IG_OOB, 0, 0);
        }
        __syncthreads();  // every wave has read its last window fragment
        park();
        __syncthreads();
    }
}

size_t igemm_relayout_floats(int KH, int KW, int Cin, int Cout)
{
    const int CoutP = (Cout + 31) / 32 * 32;
    return (size_t)KH * KW * Cin * CoutP;
}

void igemm_relayout(const float *w, int KH, int KW, int Cin, int Cout, float *out)
{
    const int CoutP = (Cout + 31) / 32 * 32, cpt = Cin / 32;
    memset(out, 0, sizeof(float) * igemm_relayout_floats(KH, KW, Cin, Cout));
    for (int tap = 0; tap < KH * KW; ++tap)
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < Cout; ++co)
                out[(((size_t)tap * cpt + ci / 32) * CoutP + co) * 32 + igemm_kpos(ci % 32)] =
                    w[((size_t)tap * Cin + ci) * Cout + co];
}

bool igemm_supported(int Cin, int Cout, int KH, int KW)
{
    return Cin > 0 && Cin % 32 == 0 && Cout > 0 && KH >= 1 && KW >= 1 && KH <= 7 && KW <= 7;
}

hipError_t launch_igemm(const float *x, int N, int H, int W, int Cin, const float *wt, int KH, int KW, int Cout,
                        int stride, int dil, const float *scale, const float *shift, const float *res, bool relu,
                        bool up2, float *y, hipStream_t s)
{
    if (!igemm_supported(Cin, Cout, KH, KW) || stride < 1 || dil < 1) return hipErrorInvalidValue;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wt = wt; a.y = y; a.scale = scale; a.shift = shift; a.res = res;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoutP = (Cout + 31) / 32 * 32;
    a.H = up2 ? 2 * H : H;  // H, W: dims of the tensor the conv SEES
    a.W = up2 ? 2 * W : W;
    a.KH = KH; a.KW = KW; a.stride = stride; a.dil = dil;
    a.relu = relu ? 1 : 0;
    a.up2 = up2 ? 1 : 0;
    // TF "SAME": out = ceil(in/stride); pad_total = max((out-1)*stride + (k-1)*dil + 1 - in, 0); before = total/2
    a.Ho = (a.H + stride - 1) / stride;
    a.Wo = (a.W + stride - 1) / stride;
    int th = (a.Ho - 1) * stride + (KH - 1) * dil + 1 - a.H; if (th < 0) th = 0;
    int tw = (a.Wo - 1) * stride + (KW - 1) * dil + 1 - a.W; if (tw < 0) tw = 0;
    a.pad_t = th / 2;
    a.pad_l = tw / 2;
    a.M = (long)N * a.Ho * a.Wo;
    if (KH == 1 && KW == 1 && Cin == 128 && Cout <= 32 && stride == 1 && up2 && !res &&
        (long)4 * H * W * Cout * 4 < (1L << 31) && (long)H * W * 512 < (1L << 31)) {  // 32-bit offsets inside one image
        ClsUpArgs q;
        q.x = x; q.wt = wt; q.scale = scale; q.shift = shift; q.y = y;
        q.N = N; q.Hs = H; q.Ws = W; q.Cout = Cout; q.relu = relu ? 1 : 0;
        q.tiles_x = cdiv_i(2 * W, HT_W); q.tiles_y = cdiv_i(2 * H, HT_H);
        const long grid = (long)N * q.tiles_x * q.tiles_y;
        if (grid < (1L << 31)) {
            ProfScope prof("k_conv1x1_up2_c128", 2.0 * (double)a.M * Cin * Cout,
                           4.0 * ((double)N * H * W * Cin + (double)a.M * Cout + (double)Cin * Cout), s);
            const long slots = 768 / launch_concurrency();  // persistent: three workgroups per CU over the chains that run side by side
            hipLaunchKernelGGL(k_conv1x1_up2_c128, dim3((unsigned)(grid < slots ? grid : slots)), dim3(256), 0, s, q);
            return hipGetLastError();
        }
    }
    if (KH == 3 && KW == 3 && Cin == 32 && stride == 1 && dil == 1 && !up2 && (long)H * W * (Cout > 32 ? Cout : 32) * 4 < (1L << 31)) {
        HaloArgs q;
        q.x = x; q.wt = wt; q.scale = scale; q.shift = shift; q.res = res; q.y = y;
        q.N = N; q.H = H; q.W = W; q.Cout = Cout; q.CoutP = a.CoutP; q.relu = relu ? 1 : 0;
        q.tiles_x = cdiv_i(W, HT_W); q.tiles_y = cdiv_i(H, HT_H); q.tiles_n = a.CoutP / 32;
        const long ntiles = (long)N * q.tiles_x * q.tiles_y * q.tiles_n;
        // persistent workgroups: two per CU (67 KB of LDS each), a multiple of the column-tile count
        long grid = (512 / launch_concurrency()) / q.tiles_n * q.tiles_n;
        if (grid < q.tiles_n) grid = q.tiles_n;
        if (grid > ntiles) grid = ntiles;
        if (ntiles < (1L << 31)) {
            ProfScope prof("k_conv3x3_c32", 2.0 * (double)a.M * 9 * Cin * Cout,
                           4.0 * ((double)N * H * W * Cin + (double)a.M * Cout * (res ? 2.0 : 1.0) + 9.0 * Cin * Cout), s);
            hipLaunchKernelGGL(k_conv3x3_c32, dim3((unsigned)grid), dim3(256), 0, s, q);
            return hipGetLastError();
        }
    }
    // 32-bit byte offsets inside the kernel: every tensor of one launch must stay below 4 GiB (split the batch); the
    // margin covers the per-chunk scalar offset (< 4 * Cin <= 8 KiB) that is added to a row offset by the hardware
    if (Cin > 2048 || (long)N * H * W * Cin * 4 >= (1L << 32) - 16384 || a.M * Cout * 4 >= (1L << 32) - 256 || a.M >= (1L << 31) - 256)
        return hipErrorInvalidValue;
    a.tiles_m = cdiv_i(a.M, IG_BM);
    const int nb32 = a.CoutP / 32;
    // widest column tile that divides the padded channel count and still leaves >= 2 workgroups per CU
    int NT = nb32 % 4 == 0 ? 4 : (nb32 % 2 == 0 ? 2 : 1);
    while (NT > 1 && (long)a.tiles_m * (nb32 / NT) < 512 / launch_concurrency()) NT >>= 1;
    a.tiles_n = nb32 / NT;
    a.ntiles = a.tiles_m * a.tiles_n;
    a.xcd_chunk = (a.ntiles + 7) / 8;
    const int grid = a.xcd_chunk * 8;
    a.trace = (g_trace_buf && (long)grid * 4 * 16 * 8 <= g_trace_bytes) ? g_trace_buf : nullptr;
#ifdef SSAL_MEASURE
    a.ablate = knobs().ablate;
#endif
    const double flops = 2.0 * (double)a.M * KH * KW * Cin * Cout;
    const double bytes = 4.0 * ((double)N * H * W * Cin + (double)a.M * Cout * (res ? 2.0 : 1.0) +
                                (double)KH * KW * Cin * Cout);
    // one profile row per kernel SYMBOL (rocprofv3 lists k_igemm<NT, false> and k_igemm<NT, true> separately)
    ProfScope prof(up2 ? (NT == 4 ? "k_igemm<4,up2>" : NT == 2 ? "k_igemm<2,up2>" : "k_igemm<1,up2>")
                       : (NT == 4 ? "k_igemm<4>" : NT == 2 ? "k_igemm<2>" : "k_igemm<1>"), flops, bytes, s);
    // the four-adjacent-pixels loader of the up-sampling form (k_igemm<.., 2>): see the kernel's header
    const bool adj = up2 && stride == 1 && a.Wo % 4 == 0 && dil % 2 == 0 && a.pad_l % 2 == 0;
    const bool sb = adj && NT >= 2 && knobs().ig_sb != 0;  // one LDS buffer, three workgroups per CU (see the kernel's header)
    const bool sb2 = !up2 && NT == 2 && knobs().ig_sb >= 2;  // the plain form at NT = 2 (27.6 KB: +0.35 % on the one-chain pass) and, from ig_sb = 3, at NT = 1 (23 KB: +0.8 %); NT = 4: -0.6 %, not used
#define SSAL_IG(N_)                                                                         \
    if (sb) hipLaunchKernelGGL((k_igemm<(N_ < 2 ? 2 : N_), 2, false, true>), dim3(grid), dim3(256), 0, s, a);   \
    else if (sb2) hipLaunchKernelGGL((k_igemm<2, 0, false, true>), dim3(grid), dim3(256), 0, s, a);   \
    else if (!up2 && NT == 1 && knobs().ig_sb >= 3) hipLaunchKernelGGL((k_igemm<1, 0, false, true>), dim3(grid), dim3(256), 0, s, a);   \
    else if (adj) hipLaunchKernelGGL((k_igemm<N_, 2>), dim3(grid), dim3(256), 0, s, a);    \
    else if (up2) hipLaunchKernelGGL((k_igemm<N_, 1>), dim3(grid), dim3(256), 0, s, a);    \
    else hipLaunchKernelGGL((k_igemm<N_, 0>), dim3(grid), dim3(256), 0, s, a)
    if (NT == 4) { SSAL_IG(4); }
    else if (NT == 2) { SSAL_IG(2); }
    else { SSAL_IG(1); }
#undef SSAL_IG
    return hipGetLastError();
}

hipError_t launch_igemm_dual(const float *x, int N, int H, int W, int Cin, const float *wt, int Cout, const float *scale,
                             const float *shift, const float *xs, int Cin_s, int stride_s, const float *wt_s,
                             const float *scale_s, const float *shift_s, bool relu, float *y, hipStream_t s)
{
    if (!igemm_supported(Cin, Cout, 1, 1) || !igemm_supported(Cin_s, Cout, 1, 1) || stride_s < 1) return hipErrorInvalidValue;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wt = wt; a.y = y; a.scale = scale; a.shift = shift; a.res = nullptr;
    a.x2 = xs; a.wt2 = wt_s; a.scale2 = scale_s; a.shift2 = shift_s;
    a.Cin2 = Cin_s; a.H2 = H * stride_s; a.W2 = W * stride_s; a.stride2 = stride_s;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.CoutP = (Cout + 31) / 32 * 32;
    a.H = H; a.W = W; a.KH = 1; a.KW = 1; a.stride = 1; a.dil = 1; a.relu = relu ? 1 : 0;
    a.Ho = H; a.Wo = W; a.M = (long)N * H * W;
    const int cmax = Cin > Cin_s ? Cin : Cin_s;
    if (cmax > 2048 || (long)N * a.H2 * a.W2 * Cin_s * 4 >= (1L << 32) - 16384 || (long)N * H * W * Cin * 4 >= (1L << 32) - 16384 ||
        a.M * Cout * 4 >= (1L << 32) - 256 || a.M >= (1L << 31) - 256)
        return hipErrorInvalidValue;
    a.tiles_m = cdiv_i(a.M, IG_BM);
    const int nb32 = a.CoutP / 32;
    int NT = nb32 % 4 == 0 ? 4 : (nb32 % 2 == 0 ? 2 : 1);
    while (NT > 1 && (long)a.tiles_m * (nb32 / NT) < 512 / launch_concurrency()) NT >>= 1;
    a.tiles_n = nb32 / NT;
    a.ntiles = a.tiles_m * a.tiles_n;
    a.xcd_chunk = (a.ntiles + 7) / 8;
    const int grid = a.xcd_chunk * 8;
    const double flops = 2.0 * (double)a.M * (Cin + Cin_s) * Cout;
    const double bytes = 4.0 * ((double)N * H * W * Cin + (double)N * H * W * Cin_s + (double)a.M * Cout + (double)(Cin + Cin_s) * Cout);
    ProfScope prof(NT == 4 ? "k_igemm<4,dual>" : NT == 2 ? "k_igemm<2,dual>" : "k_igemm<1,dual>", flops, bytes, s);
    if (NT == 4) hipLaunchKernelGGL((k_igemm<4, 0, true>), dim3(grid), dim3(256), 0, s, a);
    else if (NT == 2) hipLaunchKernelGGL((k_igemm<2, 0, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_igemm<1, 0, true>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// First convolution of a branch (ICNET_SPEC conv1_1_3x3_s2 on data_sub2, conv1_sub1 on the image):
// 3x3 / stride 2 / SAME, CIN in {1,3,4} -> 32 channels, folded BN, ReLU.  One thread per output pixel, the 27 x 32
// kernel taps are wave-uniform (scalar loads).  `sub` = 2 reads pixel (2y, 2x) of the image for conv-input pixel
// (y, x): resize_bilinear(x, H/2, W/2) under the legacy mapping is exactly that pixel (lerp weights 0).
// HBM-bound: reads the image once, writes 128 B per output pixel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_unit_f(float v) { return v; }
__device__ __forceinline__ float to_unit_f(uint8_t v) { return (float)v * (1.0f / 255.0f); }

// Workgroup = 8 x 32 output pixels of one image.  The 17 x 65 x CIN input window is staged once through LDS with
// coalesced loads (uint8 frames are converted on the way); every thread then owns one output pixel: 27 LDS reads
// against 27 x 32 FMAs whose kernel taps are wave-uniform (scalar loads).  The 32 output channels of a pixel are
// 128 contiguous bytes, but a pixel per lane would make every store instruction touch 64 different lines, so the tile
// goes back through LDS and leaves as 1-KB contiguous float4 stores.
constexpr int CF_TH = 8, CF_TW = 32, CF_IH = 2 * CF_TH + 1, CF_IW = 2 * CF_TW + 1, CF_OS = 36;

template <int CIN, typename TX>
__global__ __launch_bounds__(256) void k_conv_first(const TX *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    float *__restrict__ y, int N, int H, int W, int sub, int tiles_x,
                                                    int tiles_y)
{
    constexpr int CO = 32;
    __shared__ float xin[CF_IH * CF_IW * CIN];
    __shared__ __attribute__((aligned(16))) float xout[256 * CF_OS];
    const int Hc = H / sub, Wc = W / sub;          // dims of the conv input
    const int Ho = (Hc + 1) / 2, Wo = (Wc + 1) / 2;
    int th = (Ho - 1) * 2 + 3 - Hc; if (th < 0) th = 0;
    int tw = (Wo - 1) * 2 + 3 - Wc; if (tw < 0) tw = 0;
    const int pt = th / 2, pl = tw / 2;
    const int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * CF_TW;
    const int ty0 = ((tile / tiles_x) % tiles_y) * CF_TH;
    const long n = tile / (tiles_x * tiles_y);
    const int tid = threadIdx.x;

    // ---- stage the input window (zeros outside the conv input: SAME padding) ----------------------
    // all loads of the window are issued before the first one is consumed (unconditional loads from a clamped
    // address, zero selected afterwards): a load-then-store loop would pay one memory round trip per iteration
    const int iy0 = 2 * ty0 - pt, ix0 = 2 * tx0 - pl;
    constexpr int E = CF_IH * CF_IW * CIN, IT = (E + 255) / 256;
    float stage[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = min(tid + 256 * it, E - 1);
        const int ci = e % CIN;
        const int rx = (e / CIN) % CF_IW;
        const int ry = e / (CIN * CF_IW);
        const int iy = iy0 + ry, ix = ix0 + rx;
        const bool ok = iy >= 0 && iy < Hc && ix >= 0 && ix < Wc;
        const int iyc = min(max(iy, 0), Hc - 1), ixc = min(max(ix, 0), Wc - 1);
        const float v = to_unit_f(x[((n * H + (long)iyc * sub) * W + (long)ixc * sub) * CIN + ci]);
        stage[it] = ok ? v : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it)
        if (tid + 256 * it < E) xin[tid + 256 * it] = stage[it];
    __syncthreads();

    const int lx = tid & 31, ly = tid >> 5;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float xv = xin[((2 * ly + kh) * CF_IW + 2 * lx + kw) * CIN + ci];
                const float *wr = w + ((kh * 3 + kw) * CIN + ci) * CO;
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
            }
    // ---- folded batch-norm + ReLU, then through LDS so that the stores are contiguous --------------
#pragma unroll
    for (int q = 0; q < CO / 4; ++q) {
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = fmaf(acc[4 * q + k], scale[4 * q + k], shift[4 * q + k]);
            o[k] = v > 0.0f ? v : 0.0f;
        }
        *reinterpret_cast<float4 *>(xout + tid * CF_OS + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    // thread -> (tile row = tid >> 5 (+ 8 rows per pass? no: 256 float4 per pass = 32 pixels = one tile row))
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = j * 256 + tid;         // float4 index inside the tile: [8 rows][32 px][8 quads]
        const int q = f & 7, px = (f >> 3) & 31, row = f >> 8;
        const int oy = ty0 + row, ox = tx0 + px;
        if (oy < Ho && ox < Wo) {
            const float4 v = *reinterpret_cast<const float4 *>(xout + (row * 32 + px) * CF_OS + 4 * q);
            *reinterpret_cast<float4 *>(y + ((n * Ho + oy) * (long)Wo + ox) * CO + 4 * q) = v;
        }
    }
}

hipError_t launch_conv_first(const void *x, bool x_is_u8, int N, int H, int W, int Cin, int sub, const float *w,
                             const float *scale, const float *shift, float *y, hipStream_t s)
{
    if (sub != 1 && sub != 2) return hipErrorInvalidValue;
    const int Hc = H / sub, Wc = W / sub;
    const int Ho = (Hc + 1) / 2, Wo = (Wc + 1) / 2;
    const long total = (long)N * Ho * Wo;
    const int tiles_x = cdiv_i(Wo, CF_TW), tiles_y = cdiv_i(Ho, CF_TH);
    const long grid = (long)N * tiles_x * tiles_y;
    if (grid >= (1L << 31)) return hipErrorInvalidValue;
    ProfScope prof("k_conv_first", 2.0 * (double)total * 9 * Cin * 32,
                   (double)N * H * W * Cin * (x_is_u8 ? 1.0 : 4.0) / sub + 4.0 * (double)total * 32, s);
#define SSAL_CF(C)                                                                                                   \
    case C:                                                                                                          \
        if (x_is_u8)                                                                                                 \
            hipLaunchKernelGGL((k_conv_first<C, uint8_t>), dim3((unsigned)grid), dim3(256), 0, s, (const uint8_t *)x, w, \
                               scale, shift, y, N, H, W, sub, tiles_x, tiles_y);                                     \
        else                                                                                                         \
            hipLaunchKernelGGL((k_conv_first<C, float>), dim3((unsigned)grid), dim3(256), 0, s, (const float *)x, w,  \
                               scale, shift, y, N, H, W, sub, tiles_x, tiles_y);                                     \
        break;
    switch (Cin) {
        SSAL_CF(1) SSAL_CF(3) SSAL_CF(4)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_CF
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// tf.nn.max_pool(ksize 3x3, strides 2, "SAME") (ICNET_SPEC pool1_3x3_s2); one float4 of channels per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool3x3_s2(const float4 *__restrict__ x, int N, int H, int W, int C4,
                                                       int Ho, int Wo, int pt, int pl, float4 *__restrict__ y)
{
    const long total = (long)N * Ho * Wo * C4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C4);
        const long pix = o / C4;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const long n = pix / ((long)Wo * Ho);
        float4 best = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = 2 * oy - pt + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = 2 * ox - pl + dx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = x[((n * H + iy) * W + ix) * C4 + c];
                if (v.x > best.x) best.x = v.x;
                if (v.y > best.y) best.y = v.y;
                if (v.z > best.z) best.z = v.z;
                if (v.w > best.w) best.w = v.w;
            }
        }
        y[o] = best;
    }
}

hipError_t launch_maxpool3x3_s2(const float *x, int N, int H, int W, int C, float *y, hipStream_t s)
{
    if (C % 4) return hipErrorInvalidValue;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int th = (Ho - 1) * 2 + 3 - H; if (th < 0) th = 0;
    int tw = (Wo - 1) * 2 + 3 - W; if (tw < 0) tw = 0;
    const long total = (long)N * Ho * Wo * (C / 4);
    int grid = cdiv_i(total, 256);
    if (grid > 1 << 20) grid = 1 << 20;
    ProfScope prof("k_maxpool3x3_s2", 0.0, 4.0 * ((double)N * H * W * C + (double)N * Ho * Wo * C), s);
    hipLaunchKernelGGL(k_maxpool3x3_s2, dim3(grid), dim3(256), 0, s, (const float4 *)x, N, H, W, C / 4, Ho, Wo,
                       th / 2, tw / 2, (float4 *)y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Pyramid pooling (ICNET_SPEC conv5_3_pool{1,2,3,6}, conv5_3_sum).  Bins: b = 1 -> slot 0, b = 2 -> 1..4,
// b = 3 -> 5..13, b = 6 -> 14..49.  Bin average = (sum over the bin's rows of (sum over the bin's columns)) / count,
// both sums fp32 in ascending order (the oracle's order): k_ppm_rowsum forms the column sums of every row for the 12
// column ranges, k_ppm_pool adds the rows of each bin -- two short, wide launches instead of one thread walking a whole
// 32 x 64 map.  k_ppm_sum: y = ((((x + up1) + up2) + up3) + up6), each up_b the
// legacy-mapped bilinear resize of the b x b bin grid back to H x W.
// ------------------------------------------------------------------------------------------------
__device__ __host__ constexpr int ppm_bins(int k) { return k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 3 : 6; }
__device__ __host__ constexpr int ppm_base(int k) { return k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 5 : 14; }
constexpr int PPM_SLOTS = 50;

// column-bin slots of one row: b = 1 -> 0, b = 2 -> 1..2, b = 3 -> 3..5, b = 6 -> 6..11
__device__ __host__ constexpr int ppm_cbase(int k) { return k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 3 : 6; }
constexpr int PPM_CSLOTS = 12;

// stage 1: rowsum[n][y][cslot][c] = sum over the slot's columns (ascending) of x[n][y][.][c].  One thread per (n, y, channel
// quad) walks the row ONCE and feeds all 12 slots (a column belongs to one slot per pyramid level; neighbouring bins may share
// a column when W % b != 0): the map is read once instead of once per level, eight 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void k_ppm_rowsum(const float4 *__restrict__ x, int N, int H, int W, int C4,
                                                    float4 *__restrict__ rowsum)
{
    const long total = (long)N * H * C4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C4);
        const long ny = o / C4;  // n*H + y
        float4 s[PPM_CSLOTS];
        int x0[PPM_CSLOTS], x1[PPM_CSLOTS];
#pragma unroll
        for (int cs = 0; cs < PPM_CSLOTS; ++cs) {
            const int k = cs >= 6 ? 3 : cs >= 3 ? 2 : cs >= 1 ? 1 : 0;
            const int b = ppm_bins(k), j = cs - ppm_cbase(k);
            x0[cs] = (j * W) / b;
            x1[cs] = ((j + 1) * W + b - 1) / b;
            s[cs] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 *row = x + ny * W * C4 + c;
        for (int xb = 0; xb < W; xb += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[(long)min(xb + u, W - 1) * C4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int xx = xb + u;
#pragma unroll
                for (int cs = 0; cs < PPM_CSLOTS; ++cs) {
                    const bool in = xx >= x0[cs] && xx < x1[cs];  // xx >= W belongs to no slot
                    s[cs].x = in ? s[cs].x + v[u].x : s[cs].x;
                    s[cs].y = in ? s[cs].y + v[u].y : s[cs].y;
                    s[cs].z = in ? s[cs].z + v[u].z : s[cs].z;
                    s[cs].w = in ? s[cs].w + v[u].w : s[cs].w;
                }
            }
        }
#pragma unroll
        for (int cs = 0; cs < PPM_CSLOTS; ++cs) rowsum[(ny * PPM_CSLOTS + cs) * C4 + c] = s[cs];
    }
}

// stage 2: pooled[n][slot][c] = (sum over the bin's rows (ascending) of its row sums) / count
__global__ __launch_bounds__(256) void k_ppm_pool(const float4 *__restrict__ rowsum, int N, int H, int W, int C4,
                                                  float4 *__restrict__ pooled)
{
    const long total = (long)N * PPM_SLOTS * C4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C4);
        const int slot = (int)((o / C4) % PPM_SLOTS);
        const long n = o / ((long)C4 * PPM_SLOTS);
        const int k = slot >= 14 ? 3 : slot >= 5 ? 2 : slot >= 1 ? 1 : 0;
        const int b = ppm_bins(k), idx = slot - ppm_base(k);
        const int i = idx / b, j = idx % b;
        const int y0 = (i * H) / b, y1 = ((i + 1) * H + b - 1) / b;
        const int x0 = (j * W) / b, x1 = ((j + 1) * W + b - 1) / b;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int yy = y0; yy < y1; ++yy) {
            const float4 v = rowsum[((n * H + yy) * PPM_CSLOTS + ppm_cbase(k) + j) * C4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float cnt = (float)((y1 - y0) * (x1 - x0));
        pooled[o] = make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt);
    }
}

// One thread per (n, column, channel quad) walks the column: the horizontal half of every level's interpolation
// (top = tl + (tr - tl) * lx per bin row: 1 + 2 + 3 + 6 values) depends on the column only and is formed once -- 24 loads
// of pooled values per 32 pixels instead of 16 per pixel -- and a pixel selects its two bin rows from registers.  Same
// formula, same operand order as the per-pixel form (and the oracle): y = ((((x + up1) + up2) + up3) + up6).
template <int B> struct PpmLevel {  // one pyramid level with B x B bins: its B horizontally interpolated bin rows
    float4 hor[B];
    float hs;
    __device__ __forceinline__ void init(const float4 *__restrict__ pb, int C4, int ox, int H, int W)
    {
        const float ws = (float)B / (float)W;
        const float fx = (float)ox * ws;
        const int x0 = (int)floorf(fx);
        const int x1 = min(x0 + 1, B - 1);
        const float lx = fx - (float)x0;
        hs = (float)B / (float)H;
#pragma unroll
        for (int yb = 0; yb < B; ++yb) {
            const float4 tl = pb[(yb * B + x0) * C4], tr = pb[(yb * B + x1) * C4];
            hor[yb] = make_float4(tl.x + (tr.x - tl.x) * lx, tl.y + (tr.y - tl.y) * lx, tl.z + (tr.z - tl.z) * lx,
                                  tl.w + (tr.w - tl.w) * lx);
        }
    }
    __device__ __forceinline__ void add(float4 &r, int oy) const
    {
        const float fy = (float)oy * hs;
        const int y0 = (int)floorf(fy);
        const int y1 = min(y0 + 1, B - 1);
        const float ly = fy - (float)y0;
        float4 top = hor[0], bot = hor[0];
#pragma unroll
        for (int yb = 1; yb < B; ++yb) {
            const bool st = y0 == yb, sb = y1 == yb;  // component-wise: a ternary over float4 lvalues would select ADDRESSES
            top.x = st ? hor[yb].x : top.x; top.y = st ? hor[yb].y : top.y; top.z = st ? hor[yb].z : top.z; top.w = st ? hor[yb].w : top.w;
            bot.x = sb ? hor[yb].x : bot.x; bot.y = sb ? hor[yb].y : bot.y; bot.z = sb ? hor[yb].z : bot.z; bot.w = sb ? hor[yb].w : bot.w;
        }
        r.x = r.x + (top.x + (bot.x - top.x) * ly);
        r.y = r.y + (top.y + (bot.y - top.y) * ly);
        r.z = r.z + (top.z + (bot.z - top.z) * ly);
        r.w = r.w + (top.w + (bot.w - top.w) * ly);
    }
};

__global__ __launch_bounds__(256) void k_ppm_sum(const float4 *__restrict__ x, const float4 *__restrict__ pooled,
                                                 int N, int H, int W, int C4, float4 *__restrict__ y)
{
    const long total = (long)N * W * C4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int c = (int)(o % C4);
        const int ox = (int)((o / C4) % W);
        const long n = o / ((long)C4 * W);
        const float4 *pn = pooled + n * PPM_SLOTS * C4 + c;
        PpmLevel<1> l1;
        PpmLevel<2> l2;
        PpmLevel<3> l3;
        PpmLevel<6> l6;
        l1.init(pn + (long)ppm_base(0) * C4, C4, ox, H, W);
        l2.init(pn + (long)ppm_base(1) * C4, C4, ox, H, W);
        l3.init(pn + (long)ppm_base(2) * C4, C4, ox, H, W);
        l6.init(pn + (long)ppm_base(3) * C4, C4, ox, H, W);
        const long col = (n * H * W + ox) * C4 + c;
        for (int yb4 = 0; yb4 < H; yb4 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x[col + (long)min(yb4 + u, H - 1) * W * C4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int oy = yb4 + u;
                if (oy < H) {
                    float4 r = v[u];
                    l1.add(r, oy);
                    l2.add(r, oy);
                    l3.add(r, oy);
                    l6.add(r, oy);
                    y[col + (long)oy * W * C4] = r;
                }
            }
        }
    }
}

// how many launches of one layer the caller runs side by side (the image-group chains of the score path): the "fill the
// chip" rules below -- the widest column tile that still leaves >= 512 workgroups, 512 persistent workgroups -- apply to
// what is on the chip together, so each launch gets 512 / concurrency.  Thread-local: set around run_trunk by the caller.
static thread_local int t_concurrency = 1;
void set_launch_concurrency(int g) { t_concurrency = g < 1 ? 1 : g; }
int launch_concurrency() { return knobs().ig_div > 0 ? knobs().ig_div : t_concurrency; }

int64_t ppm_scratch_floats(int N, int H, int C) { return (int64_t)N * (PPM_SLOTS + (int64_t)PPM_CSLOTS * H) * C; }

hipError_t launch_ppm(const float *x, int N, int H, int W, int C, float *scratch, float *y, hipStream_t s)
{
    if (C % 4) return hipErrorInvalidValue;
    float *pooled = scratch;                               // [N][50][C]
    float *rowsum = scratch + (int64_t)N * PPM_SLOTS * C;  // [N][H][12][C]
    {
        const long total = (long)N * H * (C / 4);
        ProfScope prof("k_ppm_rowsum", 0.0, 4.0 * ((double)N * H * W * C + (double)total * PPM_CSLOTS * 4), s);
        hipLaunchKernelGGL(k_ppm_rowsum, dim3(cdiv_i(total, 256)), dim3(256), 0, s, (const float4 *)x, N, H, W, C / 4,
                           (float4 *)rowsum);
    }
    {
        const long total = (long)N * PPM_SLOTS * (C / 4);
        ProfScope prof("k_ppm_pool", 0.0, 4.0 * 4.0 * (double)N * H * PPM_CSLOTS * C, s);
        hipLaunchKernelGGL(k_ppm_pool, dim3(cdiv_i(total, 256)), dim3(256), 0, s, (const float4 *)rowsum, N, H, W, C / 4,
                           (float4 *)pooled);
    }
    const long total = (long)N * W * (C / 4);
    int grid = cdiv_i(total, 256);
    if (grid > 1 << 20) grid = 1 << 20;
    ProfScope prof("k_ppm_sum", 0.0, 8.0 * (double)N * H * W * C, s);
    hipLaunchKernelGGL(k_ppm_sum, dim3(grid), dim3(256), 0, s, (const float4 *)x, (const float4 *)pooled, N, H, W,
                       C / 4, (float4 *)y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv6_interp + acquisition score.  lq = logits at 1/4 resolution [N,H,W,K]; the full-resolution logits are
// resize_bilinear(lq, 4H, 4W) (legacy mapping src = dst * 0.25: y0 = oy >> 2, ly = (oy & 3) / 4).  One thread per
// 1/4-resolution pixel = one 4x4 block of output pixels: it loads its four corner logit vectors once, forms
// top/bottom rows per dx, the 16 interpolated logit vectors, and scores each in registers.  Outputs (optional):
// label / mask uint8, confidence fp32 at [N,4H,4W]; always: one fp64 partial sum per block.
// grid = (ceil(H*W/256), N): a block never straddles two images.
// ------------------------------------------------------------------------------------------------
template <int K, bool OUT>
__global__ __launch_bounds__(256) void k_upscore(const float *__restrict__ lq, int H, int W, int measure,
                                                 float threshold, double *__restrict__ partial,
                                                 uint8_t *__restrict__ label, uint8_t *__restrict__ mask,
                                                 float *__restrict__ conf)
{
    __shared__ double red[4];
    const int n = blockIdx.y;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const float inv_logK = 1.0f / logf((float)K);
    double local = 0.0;
    if (!OUT && q < (long)H * W) {
        // score only: the interpolation and the softmax front on class pairs (packed fp32: two operations per lane and
        // issue slot); element for element the operations of the generic body below, in its order
        constexpr int KP = (K + 1) / 2;
        typedef score_f32x2 f2;
        const int qx = (int)(q % W), qy = (int)(q / W);
        const int qx1 = min(qx + 1, W - 1), qy1 = min(qy + 1, H - 1);
        const float *img = lq + (long)n * H * W * K;
        f2 tl2[KP], dt2[KP], bl2[KP], db2[KP];
        {
            const float *ptl = img + ((long)qy * W + qx) * K, *ptr_ = img + ((long)qy * W + qx1) * K;
            const float *pbl = img + ((long)qy1 * W + qx) * K, *pbr = img + ((long)qy1 * W + qx1) * K;
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const int k0 = 2 * p, k1 = min(2 * p + 1, K - 1);
                tl2[p] = (f2){ptl[k0], ptl[k1]};
                bl2[p] = (f2){pbl[k0], pbl[k1]};
                dt2[p] = (f2){ptr_[k0], ptr_[k1]} - tl2[p];
                db2[p] = (f2){pbr[k0], pbr[k1]} - bl2[p];
            }
        }
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const float lx = 0.25f * (float)dx;
            const f2 lx2 = {lx, lx};
            f2 top2[KP], d2[KP];
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                top2[p] = tl2[p] + dt2[p] * lx2;
                d2[p] = (bl2[p] + db2[p] * lx2) - top2[p];
            }
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float ly = 0.25f * (float)dy;
                const f2 ly2 = {ly, ly};
                f2 l2[KP];
#pragma unroll
                for (int p = 0; p < KP; ++p) l2[p] = top2[p] + d2[p] * ly2;
                local += (double)pixel_score_only_pk<K>(l2, measure, inv_logK);
            }
        }
    } else if (q < (long)H * W) {
        const int qx = (int)(q % W), qy = (int)(q / W);
        const int qx1 = min(qx + 1, W - 1), qy1 = min(qy + 1, H - 1);
        const float *img = lq + (long)n * H * W * K;
        float tl[K], tr[K], bl[K], br[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            tl[k] = img[((long)qy * W + qx) * K + k];
            tr[k] = img[((long)qy * W + qx1) * K + k];
            bl[k] = img[((long)qy1 * W + qx) * K + k];
            br[k] = img[((long)qy1 * W + qx1) * K + k];
        }
        const int OW = 4 * W;
        const long obase = (long)n * 16 * H * W + (long)(4 * qy) * OW + 4 * qx;
        unsigned lab4[4] = {0u, 0u, 0u, 0u}, msk4[4] = {0u, 0u, 0u, 0u};
        float cf4[4][4];
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const float lx = 0.25f * (float)dx;
            float top[K], bot[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                top[k] = tl[k] + (tr[k] - tl[k]) * lx;
                bot[k] = bl[k] + (br[k] - bl[k]) * lx;
            }
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float ly = 0.25f * (float)dy;
                float l[K];
#pragma unroll
                for (int k = 0; k < K; ++k) l[k] = top[k] + (bot[k] - top[k]) * ly;
                int lab = 0;
                const float cf = OUT ? pixel_score<K>(l, measure, inv_logK, lab) : pixel_score_only<K>(l, measure, inv_logK);
                local += (double)cf;
                if (OUT) {
                    lab4[dy] |= (unsigned)lab << (8 * dx);
                    msk4[dy] |= (cf < threshold ? 0u : 1u) << (8 * dx);
                    cf4[dy][dx] = cf;
                }
            }
        }
        if (OUT) {
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const long o = obase + (long)dy * OW;
                if (label) *reinterpret_cast<unsigned *>(label + o) = lab4[dy];
                if (mask) *reinterpret_cast<unsigned *>(mask + o) = msk4[dy];
                if (conf) *reinterpret_cast<float4 *>(conf + o) = make_float4(cf4[dy][0], cf4[dy][1], cf4[dy][2], cf4[dy][3]);
            }
        }
    }
    const double rsum = block_sum_256(local, red);
    if (threadIdx.x == 0) partial[(long)n * gridDim.x + blockIdx.x] = rsum;
}

int upscore_blocks(int H, int W) { return cdiv_i((long)H * W, 256); }

hipError_t launch_upscore(const float *lq, int N, int H, int W, int K, int measure, float threshold, double *partial,
                          uint8_t *label, uint8_t *mask, float *conf, hipStream_t s)
{
    dim3 grid(upscore_blocks(H, W), N), block(256);
    ProfScope prof("k_upscore", 16.0 * (double)N * H * W * K * 6.0,
                   4.0 * (double)N * H * W * K +
                       16.0 * (double)N * H * W * ((label ? 1 : 0) + (mask ? 1 : 0) + (conf ? 4 : 0)), s);
    const bool out = label || mask || conf;
#define SSAL_US(KK)                                                                                              \
    case KK:                                                                                                     \
        if (out)                                                                                                 \
            hipLaunchKernelGGL((k_upscore<KK, true>), grid, block, 0, s, lq, H, W, measure, threshold, partial,  \
                               label, mask, conf);                                                               \
        else                                                                                                     \
            hipLaunchKernelGGL((k_upscore<KK, false>), grid, block, 0, s, lq, H, W, measure, threshold, partial, \
                               label, mask, conf);                                                               \
        break;
    switch (K) {
        SSAL_US(2) SSAL_US(3) SSAL_US(4) SSAL_US(5) SSAL_US(6) SSAL_US(7) SSAL_US(8) SSAL_US(9) SSAL_US(10)
        SSAL_US(11) SSAL_US(12) SSAL_US(13) SSAL_US(14) SSAL_US(15) SSAL_US(16) SSAL_US(17) SSAL_US(18)
        SSAL_US(19) SSAL_US(20) SSAL_US(21) SSAL_US(22) SSAL_US(23) SSAL_US(24) SSAL_US(25) SSAL_US(26)
        SSAL_US(27) SSAL_US(28) SSAL_US(29) SSAL_US(30) SSAL_US(31) SSAL_US(32)
    default:
        return hipErrorInvalidValue;
    }
#undef SSAL_US
    return hipGetLastError();
}

}  // namespace ssal
