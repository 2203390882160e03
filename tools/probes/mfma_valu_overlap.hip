// Do fp32 MFMAs and packed fp32 VALU FMAs of DIFFERENT waves on one SIMD overlap?  (measurement aid, gfx950)
//   hipcc --offload-arch=gfx950:xnack- -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/mfma_valu_overlap && ./mfma_valu_overlap
// A workgroup = 8 waves = 2 per SIMD.  Role of a wave ((wave id >> 2) & 1, so that every SIMD gets one of each): 0 = a chain of v_mfma_f32_32x32x2_f32 (4 independent
// accumulators), 1 = chains of v_pk_fma_f32 (16 independent register pairs).  Modes: MFMA waves only, VALU waves only, both.
// If the two kinds of work ran on separate pipes, "both" would take max(a, b); if they share the fp32 FMA datapath, a + b.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int VK>  // matrix role: 0 = fp32 32x32x2, 1 = bf16 32x32x16; vector role: 0 = v_pk_fma_f32, 1 = v_fma_f32, 2 = integer add / xor, 3 = v_exp_f32
__global__ __launch_bounds__(512) void k_probe(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6, role = (wave >> 2) & 1;  // waves 0-3: one MFMA wave per SIMD, waves 4-7: one VALU wave per SIMD
    const bool run_m = role == 0 && (mode & 1), run_v = role == 1 && (mode & 2);
    float r = 0.0f;
    if (run_m) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        const float x = 1.0f + threadIdx.x * 1e-7f, y = 0.5f;
        bf16x8 bx, by;
        for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    if (run_v) {
        if (VK == 0) {
            f32x2 v[16];
            for (int i = 0; i < 16; ++i) v[i] = (f32x2){1.0f + i, 2.0f + threadIdx.x};
            const f32x2 m = {1.000001f, 0.999999f}, c = {1e-6f, -1e-6f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);  // 16 independent v_pk_fma_f32
            }
            for (int i = 0; i < 16; ++i) r += v[i].x + v[i].y;
        } else if (VK == 1) {
            float v[16];
            for (int i = 0; i < 16; ++i) v[i] = 1.0f + i + threadIdx.x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.000001f, 1e-6f);  // 16 independent v_fma_f32
            }
            for (int i = 0; i < 16; ++i) r += v[i];
        } else if (VK == 2) {
            unsigned v[16];
            for (int i = 0; i < 16; ++i) v[i] = i + threadIdx.x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { v[i] = (v[i] + 0x9e3779b9u) ^ (unsigned)it; asm volatile("" : "+v"(v[i])); }  // v_add + v_xor
            }
            for (int i = 0; i < 16; ++i) r += (float)v[i];
        } else {
            float v[16];
            for (int i = 0; i < 16; ++i) v[i] = 0.001f * (i + 1);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f; }  // v_exp_f32 (quarter rate) + v_add
            }
            for (int i = 0; i < 16; ++i) r += v[i];
        }
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int KIND, int VK> static float run(int mode, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<KIND, VK>), dim3(256 * 2), dim3(512), 0, 0, out, iters / 10, mode);  // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<KIND, VK>), dim3(256 * 2), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND, int VK> static void report(const char *mname, const char *vname, int iters, float *out)
{
    const float tm = run<KIND, VK>(1, iters, out), tv = run<KIND, VK>(2, iters, out), tb = run<KIND, VK>(3, iters, out);
    printf("%-14s alone %.3f ms | %-22s alone %.3f ms | both on every SIMD %.3f ms   (max %.3f, sum %.3f: overlap %.0f %%)\n", mname, tm, vname,
           tv, tb, tm > tv ? tm : tv, tm + tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main()
{
    float *out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    // a launch: 512 workgroups (2 per CU) x 4 MFMA waves x iters x 4 MFMAs  |  x 4 vector waves x iters x 16 (x 2) instructions
    report<0, 0>("fp32 32x32x2", "v_pk_fma_f32", iters, out);
    report<0, 1>("fp32 32x32x2", "v_fma_f32", iters, out);
    report<0, 2>("fp32 32x32x2", "v_add_u32 + v_xor_b32", iters, out);
    report<0, 3>("fp32 32x32x2", "v_exp_f32 + v_add_f32", iters, out);
    report<1, 0>("bf16 32x32x16", "v_pk_fma_f32", iters, out);
    report<1, 1>("bf16 32x32x16", "v_fma_f32", iters, out);
    report<1, 2>("bf16 32x32x16", "v_add_u32 + v_xor_b32", iters, out);
    return 0;
}
