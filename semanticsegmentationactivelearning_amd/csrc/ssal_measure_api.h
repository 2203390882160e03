/* ssal_measure_api.h -- extra C entry points of the MEASUREMENT libraries only (libssal_hip_measure.so,
 * libssal_hip_trace.so: the product sources + csrc/ssal_probe.hip, compiled with -DSSAL_MEASURE by
 * tools/phase_trace.py).  The product libssal_hip.so neither contains these kernels nor exports these symbols
 * (tests/test_host_cpu.py compares its export table with include/*.h).  No reference counterpart. */
#ifndef SSAL_MEASURE_API_H
#define SSAL_MEASURE_API_H
#ifdef __cplusplus
extern "C" {
#endif

/* measurement aid: bare fp32 MFMA loop (shape 32 = 32x32x2, 16 = 16x16x4), 4 waves per block, 4
 * independent accumulators per wave; out_dev needs blocks*256 floats.  Time it with ssal_profile_*. */
int ssal_debug_mfma_peak(int shape, int blocks, int iters, float *out_dev, void *stream);

/* measurement aid (tools/mem_probe.py): y = x for an [n,h,w,64] tensor with the access shape `mode`
 * (0 linear, 1 MFMA-fragment tile, 2 coalesced tile, 3/4 = 1/2 + halo-ring reads, 5-8 other tile shapes, 9 group by group, 10-13 linear with 16 / 16 / 4 / 2 float4 per thread, 14 = 10 in slab order, 15-18 persistent workgroups that prefetch the next tile); h % 8 == 0, w % 32 == 0;
 * spin = shader clocks of ALU work between the loads and the stores. */
int ssal_debug_copy_probe(int mode, const float *x_dev, float *y_dev, int n, int h, int w, int spin,
                          void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSAL_MEASURE_API_H */
