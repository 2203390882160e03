// ssal_bottleneck_args.h -- kernel argument blocks of the MFMA-fused bottleneck kernels
// (shared by the 32x32x2 family, ssal_bottleneck_mfma.hip, and the 16x16x4 family,
// ssal_bottleneck_mfma16.hip).  Channel counts in the comments are those of the 32-wide family;
// the 16-wide family uses the same fields with its own (smaller) shapes.
#pragma once
#include <stdint.h>

#include "ssal_measure.h"
#ifdef SSAL_MEASURE
#define SSAL_ABLATE_IS(args, v) ((args).ablate == (v))
#else
#define SSAL_ABLATE_IS(args, v) false
#endif

namespace ssal {

// float offsets of the three kernels inside BnkArgs::wq (each [chunk][q][lane 64][4]: element i = the fragment of MFMA step 4 q + i)
namespace quad {
constexpr int WP = 0, WC = 128 * 32;                                              // wp: 16 quads; wc: taps x 4 quads
constexpr __host__ __device__ int we(int taps) { return WC + taps * 32 * 32; }     // we: 4 N-tiles x 4 quads
constexpr __host__ __device__ int total(int taps) { return we(taps) + 32 * 128; }
}

// regular / dilated / asymmetric bottleneck (enet_modules.py:526-599)
struct BnkArgs {
    const float *x;
    float *y;
    const float *wp, *ps, *pt, *pa;  // proj kernel [128][32], folded BN, alpha
    const float *wc, *cs, *ct, *ca;  // conv kernel [3][3][32][32] (HWIO; [5][1][32][32] if asym), folded BN, alpha
    const float *wc2;                // asymmetric only: second kernel [1][5][32][32]
    const float *we, *es, *et, *ra;  // exp kernel [32][128], folded BN, residual alpha
    const float *wq;                 // regular 128-channel block: wp | wc (asymmetric: the (5,1) kernel then the (1,5) kernel) | we in QUAD layout (quad::, ssal_host.h: bnk_quad_layout)
    int N, H, W, dil;
    int TH;                // tile rows (phase space)
    int tiles_y, tiles_x;  // tiles per phase sub-image (sized for the largest phase)
    unsigned long long *trace;  // phase-trace buffer (NULL unless a -DSSAL_PHASE_TRACE build is being traced)
    int ntiles, xcd_chunk;  // XCD-aware tile order: tile = (b % 8) * xcd_chunk + b / 8 (xcd_chunk = 0: tile = b)
#ifdef SSAL_MEASURE
    int ablate;            // measurement builds only (tools/phase_trace.py): 1 = stop after the projection phase,
                           // 2 = skip the projection phase, 3 = no residual traffic, 4 = no store traffic, 5 = 3 + 4,
                           // 6 = projection loads all hit pixel 0, 7 = 5 + 6  (results invalid; timing only)
#endif
};

// downsample bottleneck (enet_modules.py:868-938)
struct DownArgs {
    const float *x;                  // [N,H,W,64]
    float *y;                        // [N,H/2,W/2,128]
    uint8_t *code;                   // [N,H/2,W/2,64]
    const float *wp, *ps, *pt, *pa;  // proj kernel [2][2][64][32], folded BN, alpha
    const float *wc, *cs, *ct, *ca;  // conv kernel [3][3][32][32]
    const float *we, *es, *et, *ra;  // exp kernel [32][128]
    int N, H, W;                     // INPUT dims (even)
    int TH, tiles_y, tiles_x;
    unsigned long long *trace;       // phase-trace buffer (NULL unless a -DSSAL_PHASE_TRACE build is being traced)
};

// upsample bottleneck (enet_modules.py:1217-1292)
struct UpArgs {
    const float *x;
    float *y;
    const uint8_t *code;             // [N,H,W,64] window codes dy*2+dx saved by the matching downsample
    const float *wp, *ps, *pt, *pa;  // proj kernel [128][32], folded BN, alpha
    const float *ws;                 // stacked transposed-conv kernel [6][32][32]
    const float *cs, *ct, *ca;       // [16]
    const float *we, *es, *et;       // exp kernel [16][64], folded BN [64]
    const float *wr;                 // residual kernel [128][64]
    const float *ra;                 // [64]
    int N, H, W, dil;                // dil == 1
    int TH, tiles_y, tiles_x;
    unsigned long long *trace;       // phase-trace buffer (NULL unless a -DSSAL_PHASE_TRACE build is being traced)
};

}  // namespace ssal
