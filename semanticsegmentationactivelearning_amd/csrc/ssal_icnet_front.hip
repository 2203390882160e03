// ssal_icnet_front.hip -- the first TWO convolutions of an ICNet branch in one launch (score path only), gfx950.
//
//   high-resolution branch (ICNET_SPEC section 3):  conv1_sub1 (3x3 / s2, image -> 32) + conv2_sub1 (3x3 / s2, 32 -> 32)
//   shared stem            (ICNET_SPEC section 1):  conv1_1_3x3_s2 (3x3 / s2 on data_sub2) + conv1_2_3x3 (3x3 / s1)
//
// Why: the first convolution is HBM-bound on its OUTPUT (128 B written per pixel for 12 B read: 537 MB for a batch of
// eight 1024 x 2048 frames on the high-resolution branch) and the second one reads all of it back.  Here a workgroup
// owns 8 x 16 pixels of the SECOND convolution's output, evaluates the first convolution on the window those pixels
// read (17 x 33 pixels for stride 2, 10 x 18 for stride 1) on the vector ALU straight into LDS, and runs the second
// convolution's 144 MFMAs per wave from there: the intermediate tensor never exists in HBM.
//
//   phase 1 (VALU)  work item = (pass of 64 window pixels, group of 8 channels): the kernel taps of an item are
//                   wave-uniform (scalar loads, SGPR pairs into v_pk_fma_f32, the activation broadcast to both halves).
//                   Stride 2: 561 window pixels = 8.8 passes x 4 channel groups = 36 items, nine per wave -- wave w
//                   owns passes 2w and 2w+1 (all four channel groups, two pixels per lane share each tap) plus channel
//                   group w of the last, 49-pixel pass: the lanes are 97 % busy (a pixel per thread with all 32
//                   channels, as k_conv_first does it, needs three passes of 256 for 561 pixels: 73 %) and a lane reads
//                   the 27 image values of only THREE pixels per tile.  Those 27 loads per lane are issued one tile
//                   ahead, right behind the barrier that opens phase 2, and land under its MFMAs.
//                   Stride 1: 180 pixels = 3 passes; wave w evaluates channel group w of all three.
//   phase 2 (MFMA)  k_conv3x3_c32's K loop on the window; the 9 x 32 x 32 kernel slice is NOT staged in LDS (41 KB: the
//                   stride-2 window's 79 KB would then leave room for one workgroup per CU) -- the fragments of the first
//                   four taps are resident in registers, the others come from L1 / L2 one tap ahead, as in ENet's fused
//                   bottleneck.  For stride 2 the window's columns are stored
//                   de-interleaved (even columns, then odd columns of a row), so that the 16 pixels a quarter-wave
//                   reads for one tap are neighbours in LDS (conflict-free ds_read_b128).
//   Two workgroups per CU: one's MFMA phase runs under the other's VALU phase.
//
// Bit-exactness: both convolutions keep the chain order of the separate launches -- conv 1: fmaf(x, w, acc) over (kh, kw,
// ci) ascending from +0 with zeros for SAME padding (k_conv_first), y = max(fmaf(acc, s, t), 0); conv 2: the fp32 MFMA
// chain over (kh, kw, ci) ascending on permuted-k rows (k_igemm / k_conv3x3_c32), window pixels outside conv 1's OUTPUT
// are exact zeros (SAME padding applies to the tensor conv 2 reads, not to conv 1 evaluated off the image).
#include "ssal_icnet.h"
#include "ssal_internal.h"
#include "ssal_mfma.h"
#include "ssal_prof.h"

namespace ssal {

namespace {

constexpr int FR_LDK = 36;            // floats per window pixel in LDS (32 + 4: conflict-free b128 rows)
constexpr int FR_TH = 8, FR_TW = 16;  // output tile of the second convolution
constexpr unsigned FR_OOB = 0xFFFFFFFFu;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(4))) float cfloat_t;  // constant address space

struct FrontArgs {
    const void *x;               // image [N,H,W,CIN], float or uint8  (conv 1's kernel HWIO [3][3][CIN][32] + folded BN: kernel parameters)
    const float *w2, *s2, *t2;   // conv 2: kernel in igemm layout [9][1][32][32 permuted-k], folded batch-norm
    float *y;                    // [N,H2,W2,32]
    int N, H, W, sub;            // conv 1 runs on image[::sub, ::sub] (sub = 2: ICNET_SPEC data_sub2)
    int H1, W1, H2, W2;          // conv 1 / conv 2 output dims
    int tiles_x, tiles_y;
};

__device__ __forceinline__ float front_unit(uint8_t v) { return (float)v * (1.0f / 255.0f); }

// the CIN values of one image pixel (byte offset `off`, FR_OOB = outside: zeros)
template <int CIN>
__device__ __forceinline__ void front_load_px(rsrc_t rs, unsigned off, const float *, float (&v)[CIN])
{
    if constexpr (CIN == 3) {
        const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
        v[0] = __uint_as_float(q[0]); v[1] = __uint_as_float(q[1]); v[2] = __uint_as_float(q[2]);
    } else if constexpr (CIN == 4) {
        const float4 q = bload4(rs, off, 0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[c] = bload(rs, off, 4 * c);
    }
}
template <int CIN>
__device__ __forceinline__ void front_load_px(rsrc_t rs, unsigned off, const uint8_t *, float (&v)[CIN])
{
#pragma unroll
    for (int c = 0; c < CIN; ++c) v[c] = front_unit((uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, off, c, 0));
}

// conv 1 for NP pixels per lane (image values v[j][tap][ci], window slots slot[j], ok[j] = inside conv 1's output) and
// the channel groups cg_lo .. cg_hi - 1 (wave-uniform): 27 x 8 taps per group stream through two SGPR buffers of one
// (kh, kw) each -- the scalar loads of tap t + 1 are issued before the FMAs of tap t, and the explicit lgkmcnt(0)
// completes tap t BEFORE that prefetch is issued (FsTap's scheme, ssal_kernels.hip).
template <int CIN, int NP, int LDK>
__device__ __forceinline__ void front_conv1(const float (*v)[9][CIN], const int *slot, const bool *ok, int cg_lo, int cg_hi,
                                            const float *w1g, const float *s1g, const float *t1g, float *Ws, int WP)
{
    constexpr int kWaitScalar = 0xC07F;  // s_waitcnt lgkmcnt(0), vmcnt / expcnt untouched
#pragma unroll 1
    for (int cg = cg_lo; cg < cg_hi; ++cg) {
        // constant address space: a wave-uniform read from it is a scalar load whatever the alias analysis concludes; the
        // opaque copy per group keeps the (loop-invariant) taps from being hoisted into spilled SGPRs
        unsigned long w1u = (unsigned long)(w1g + 8 * cg), s1u = (unsigned long)(s1g + 8 * cg), t1u = (unsigned long)(t1g + 8 * cg);
        asm volatile("" : "+s"(w1u), "+s"(s1u), "+s"(t1u));
        const cfloat_t *w1 = (const cfloat_t *)w1u, *s1 = (const cfloat_t *)s1u, *t1 = (const cfloat_t *)t1u;
        float wA[CIN * 8], wB[CIN * 8];
        auto wload = [&](int tap, float (&w)[CIN * 8]) {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int k = 0; k < 8; ++k) w[ci * 8 + k] = w1[(tap * CIN + ci) * 32 + k];
        };
        f32x2 acc[NP][4];
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[j][p] = f32x2{0.0f, 0.0f};
        wload(0, wA);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float (&wc)[CIN * 8] = (tap & 1) ? wB : wA;
            float (&wn)[CIN * 8] = (tap & 1) ? wA : wB;
            __builtin_amdgcn_s_waitcnt(kWaitScalar);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9) wload(tap + 1, wn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const f32x2 a2 = {v[j][tap][ci], v[j][tap][ci]};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[j][p] = __builtin_elementwise_fma(a2, f32x2{wc[ci * 8 + 2 * p], wc[ci * 8 + 2 * p + 1]}, acc[j][p]);
                }
            // pin this tap's FMAs HERE: the builtins around them order only side effects, and instruction selection is free
            // to line all 27 scalar loads up first (into spilled SGPRs) and the FMAs behind them
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(acc[j][p]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // folded batch-norm + ReLU -> the window (exact zeros outside conv 1's output: conv 2's SAME padding)
        float sc[8], sh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = s1[k]; sh[k] = t1[k]; }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float o[8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float v0 = fmaf(acc[j][p][0], sc[2 * p], sh[2 * p]), v1 = fmaf(acc[j][p][1], sc[2 * p + 1], sh[2 * p + 1]);
                o[2 * p] = ok[j] && v0 > 0.0f ? v0 : 0.0f;
                o[2 * p + 1] = ok[j] && v1 > 0.0f ? v1 : 0.0f;
            }
            // permuted-k row: [c0 c2 c4 c6 | c1 c3 c5 c7].  Unconditional (surplus lanes of the last pass write the spare
            // row WP): under `if (slot < WP)` the compiler sinks the pixel's FMAs into the branch and keeps every tap alive
            float *wp_ = Ws + min(slot[j], WP) * LDK + 8 * cg;
            *reinterpret_cast<float4 *>(wp_) = make_float4(o[0], o[2], o[4], o[6]);
            *reinterpret_cast<float4 *>(wp_ + 4) = make_float4(o[1], o[3], o[5], o[7]);
        }
    }
}

template <int CIN, typename TX, int S2>
__global__ __launch_bounds__(256, 2) void k_front2(FrontArgs a, const float *__restrict__ w1g,
                                                   const float *__restrict__ s1g, const float *__restrict__ t1g)
{
    constexpr int LDK = FR_LDK;
    constexpr int WH = S2 * (FR_TH - 1) + 3, WW = S2 * (FR_TW - 1) + 3;  // conv-1 window: 17 x 33 (s2) / 10 x 18 (s1)
    constexpr int NE = (WW + 1) / 2;                                     // even columns of a window row (s2 layout)
    constexpr int WP = WH * WW;
    static_assert((WP + 63) / 64 == (S2 == 2 ? 9 : 3), "pass ownership below is written for 9 / 3 passes");
    __shared__ __attribute__((aligned(16))) float Ws[(WP + 1) * LDK];  // + one row the surplus lanes of the last pass write
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int nsp = a.N * a.tiles_y * a.tiles_x;
    int sp = blockIdx.x;
    if (sp >= nsp) return;  // whole workgroup, before any barrier
    const int step = gridDim.x;

    const rsrc_t wrs = make_rsrc(a.w2, 9u * 32u * 32u * 4u);
    const unsigned blo = (unsigned)(r * 32 + 4 * h) * 4u;  // this lane's part of a conv-2 fragment address
    const float bsc = a.s2[r], bsh = a.t2[r];
    const int Hc = a.H / a.sub, Wc = a.W / a.sub;  // conv 1's input dims (even: the launcher checks)
    const unsigned img_bytes = (unsigned)((long)a.H * a.W * CIN * sizeof(TX));
    const unsigned ybytes = (unsigned)(a.H2 * a.W2 * 32 * 4);

    const int pr = 2 * wave + (r >> 4), pc = r & 15;  // this lane's output pixel inside the tile
    // first window slot this lane's fragments come from (tap (0, 0)); taps add tap_slot(kh, kw)
    const float *Ab = Ws + (S2 * pr * WW + pc) * LDK + 4 * h;
    auto tap_slot = [](int kh, int kw) { return S2 == 2 ? kh * WW + (kw == 1 ? NE : (kw >> 1)) : kh * WW + kw; };

    // the three window pixels of this lane: stride 2 -> passes 2 wave, 2 wave + 1 and the last one (8); stride 1 -> 0, 1, 2
    int slot[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) slot[j] = lane + 64 * (S2 == 2 ? (j < 2 ? 2 * wave + j : 8) : j);
    int wy[3], wx[3];  // window coordinates (row, column) of the three pixels
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int wr = slot[j] / WW, ci = slot[j] - wr * WW;
        wy[j] = wr;
        wx[j] = S2 == 2 ? (ci < NE ? 2 * ci : 2 * (ci - NE) + 1) : ci;
    }
    struct Tile { int n, ty0, tx0; };
    auto decode = [&](int t) {
        const int tx = t % a.tiles_x, q_ = t / a.tiles_x;
        return Tile{q_ / a.tiles_y, (q_ % a.tiles_y) * FR_TH, tx * FR_TW};
    };
    // pixel j of tile t: inside conv 1's output?  (y10, x10 = window origin in conv 1's output)
    auto inside = [&](const Tile &t, int j, int &y1, int &x1) {
        y1 = (S2 == 2 ? 2 * t.ty0 : t.ty0 - 1) + wy[j];
        x1 = (S2 == 2 ? 2 * t.tx0 : t.tx0 - 1) + wx[j];
        return slot[j] < WP && y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1;
    };
    // the 27 image values of each of the three pixels.  Unconditional buffer loads: outside the image / the window / (live
    // = false: the call behind the last tile) = out-of-range offset = zeros.  The call itself stays unconditional: under
    // `if (next tile)` the loaded registers become loop-carried PHIs whose copies -- and the wait for the loads -- the
    // compiler places right behind the loads, i.e. BEFORE the K loop they are meant to overlap.
    float v[3][9][CIN];
    auto load_image = [&](int t_, bool live) {
        const Tile t = decode(t_);
        const rsrc_t xrs = make_rsrc(reinterpret_cast<const TX *>(a.x) + (long)t.n * a.H * a.W * CIN, img_bytes);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int y1, x1;
            const bool ok1 = live && inside(t, j, y1, x1);
            // SAME padding of a 3x3 / s2 conv on an even size: nothing before, one row / column after
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int yy = 2 * y1 + kh, xx = 2 * x1 + kw;
                    const bool ok = ok1 && yy < Hc && xx < Wc;
                    const unsigned off = ok ? (unsigned)(((yy * a.sub) * a.W + xx * a.sub) * CIN) * (unsigned)sizeof(TX) : FR_OOB;
                    front_load_px<CIN>(xrs, off, (const TX *)nullptr, v[j][kh * 3 + kw]);
                }
        }
    };
    load_image(sp, true);
    // conv 2's kernel fragments: those of the first RES taps stay in registers for the life of the (persistent) workgroup
    // (64 of the 73 registers the 256-register budget of two workgroups per CU leaves), the others come from L1 / L2 one
    // tap ahead.  All 36 resident would be 144 registers -- what the first version of this kernel did, at the price of
    // having no room to request the image one tile ahead.  0 -> 4 resident taps: 284 -> 274 us.
    constexpr int RES = 4;
    auto load_b = [&](int tap, float4 (&b)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = bload4(wrs, blo, (unsigned)(tap * 1024 + 8 * g) * 4u);
    };
    float4 bres[RES > 0 ? RES : 1][4];
#pragma unroll
    for (int tp = 0; tp < RES; ++tp) load_b(tp, bres[tp]);

    for (; sp < nsp; sp += step) {
        const Tile t = decode(sp);
        // ---- phase 1: conv 1 on the window ----
        {
            bool ok1[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int y1, x1;
                ok1[j] = inside(t, j, y1, x1);
            }
            if (S2 == 2) {
                front_conv1<CIN, 2, LDK>(v, slot, ok1, 0, 4, w1g, s1g, t1g, Ws, WP);
                front_conv1<CIN, 1, LDK>(v + 2, slot + 2, ok1 + 2, wave, wave + 1, w1g, s1g, t1g, Ws, WP);
            } else {
                front_conv1<CIN, 3, LDK>(v, slot, ok1, wave, wave + 1, w1g, s1g, t1g, Ws, WP);
            }
        }
        __syncthreads();
        {
            const bool more = sp + step < nsp;
            load_image(more ? sp + step : sp, more);  // in flight during the K loop below
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- phase 2: conv 2, 9 taps x 4 groups of 8 channels; kernel fragments one tap ahead, A fragments one group ----
        f32x16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.0f;
        float4 af[2], bq[2][4];
        load_b(RES, bq[RES & 1]);
        af[0] = *reinterpret_cast<const float4 *>(Ab);
#pragma unroll
        for (int s = 0; s < 36; ++s) {  // s = tap * 4 + group
            const int c = s & 1, nx = c ^ 1, tap = s >> 2, g = s & 3;
            if (g == 0 && tap + 1 > RES && tap + 1 < 9) load_b(tap + 1, bq[(tap + 1) & 1]);
            if (s + 1 < 36) {
                const int t1_ = (s + 1) >> 2, g1 = (s + 1) & 3;
                af[nx] = *reinterpret_cast<const float4 *>(Ab + tap_slot(t1_ / 3, t1_ % 3) * LDK + 8 * g1);
            }
            const float4 b = tap < RES ? bres[tap][g] : bq[tap & 1][g];
            acc2 = mfma32(af[c].x, b.x, acc2);
            acc2 = mfma32(af[c].y, b.y, acc2);
            acc2 = mfma32(af[c].z, b.z, acc2);
            acc2 = mfma32(af[c].w, b.w, acc2);
        }

        // ---- epilogue: folded batch-norm + ReLU, lane = output channel ----
        const rsrc_t yrs = make_rsrc(a.y + (long)t.n * a.H2 * a.W2 * 32, ybytes);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int oy = t.ty0 + 2 * wave + (m >> 4), ox = t.tx0 + (m & 15);
            float o = fmaf(acc2[i], bsc, bsh);
            o = o > 0.0f ? o : 0.0f;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), yrs,
                                                  (oy < a.H2 && ox < a.W2) ? (unsigned)(((oy * a.W2 + ox) * 32 + r) * 4) : FR_OOB, 0, 0);
        }
        __syncthreads();  // every wave has finished reading the window
    }
}

}  // namespace

bool front2_supported(int H, int W, int Cin, int sub, int stride2)
{
    if (Cin != 1 && Cin != 3 && Cin != 4) return false;
    if (sub != 1 && sub != 2) return false;
    if (stride2 != 1 && stride2 != 2) return false;
    // every level even: SAME padding of the stride-2 convolutions is then "nothing before, one after"
    const int Hc = H / sub, Wc = W / sub;
    if (H % sub || W % sub || Hc % 2 || Wc % 2) return false;
    if (stride2 == 2 && ((Hc / 2) % 2 || (Wc / 2) % 2)) return false;
    return true;
}

hipError_t launch_front2(const void *x, bool x_is_u8, int N, int H, int W, int Cin, int sub, const float *w1,
                         const float *s1, const float *t1, const float *w2_igemm, const float *s2, const float *t2,
                         int stride2, float *y, hipStream_t s)
{
    if (!front2_supported(H, W, Cin, sub, stride2)) return hipErrorInvalidValue;
    FrontArgs a;
    a.x = x; a.w2 = w2_igemm; a.s2 = s2; a.t2 = t2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.sub = sub;
    a.H1 = H / sub / 2; a.W1 = W / sub / 2;
    a.H2 = a.H1 / stride2; a.W2 = a.W1 / stride2;
    a.tiles_x = (a.W2 + FR_TW - 1) / FR_TW;
    a.tiles_y = (a.H2 + FR_TH - 1) / FR_TH;
    const long tiles = (long)N * a.tiles_x * a.tiles_y;
    if (tiles >= (1L << 31) || (long)H * W * Cin * 4 >= (1L << 31)) return hipErrorInvalidValue;
    // persistent workgroups, two per CU over all chains that run side by side
    long grid = 512 / launch_concurrency();
    if (grid < 1) grid = 1;  // knob ig_div > 512: the launch degrades, it does not fail
    if (grid > tiles) grid = tiles;
    const double px1 = (double)N * a.H1 * a.W1, px2 = (double)N * a.H2 * a.W2;
    ProfScope prof(stride2 == 2 ? "k_front2<s2>" : "k_front2<s1>", 2.0 * px1 * 9 * Cin * 32 + 2.0 * px2 * 9 * 32 * 32,
                   (double)N * H * W * Cin * (x_is_u8 ? 1.0 : 4.0) / (sub * sub) + 4.0 * px2 * 32, s);
#define SSAL_FR(C, T, S_) hipLaunchKernelGGL((k_front2<C, T, S_>), dim3((unsigned)grid), dim3(256), 0, s, a, w1, s1, t1)
#define SSAL_FR_T(C)                                                              \
    case C:                                                                       \
        if (x_is_u8) { if (stride2 == 2) SSAL_FR(C, uint8_t, 2); else SSAL_FR(C, uint8_t, 1); } \
        else { if (stride2 == 2) SSAL_FR(C, float, 2); else SSAL_FR(C, float, 1); }             \
        break
    switch (Cin) {
        SSAL_FR_T(1);
        SSAL_FR_T(3);
        SSAL_FR_T(4);
    default: return hipErrorInvalidValue;
    }
#undef SSAL_FR_T
#undef SSAL_FR
    return hipGetLastError();
}

}  // namespace ssal
