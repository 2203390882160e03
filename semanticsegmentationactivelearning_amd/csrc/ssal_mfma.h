// ssal_mfma.h -- gfx950 fp32-input MFMA helpers shared by the fused bottleneck kernels.
//
// v_mfma_f32_32x32x2_f32  (l = lane, r = l & 31, h = l >> 5):
//   A: lane holds A[row r][k = h]    B: lane holds B[k = h][col r]
//   D: reg i of lane holds D[row (i&3) + 8*(i>>2) + 4*h][col r]          (16 registers)
// v_mfma_f32_16x16x4_f32  (i = l & 15, g = l >> 4):
//   A: A[row i][k = g]               B: B[k = g][col i]
//   D: reg r holds D[row 4*g + r][col i]                                  (4 registers)
// Both are exact fp32: the result is a k-ordered fmaf chain (k = 0.. within an instruction, then
// instruction order), which is what lets the kernels be bit-identical to the parity oracle as long
// as every operand is fed in ascending (kh, kw, ci) order.  The permlane helpers below re-arrange
// registers that hold consecutive channels per lane into that order without touching LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace ssal {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// after the call: a = [a.lo | b.lo], b = [a.hi | b.hi]   (lo = lanes 0-31, hi = lanes 32-63)
__device__ __forceinline__ void swap32(float &a, float &b)
{
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// after the call (q0..q3 = the four 16-lane quarters):
//   a = [a.q0 b.q0 a.q2 b.q2], b = [a.q1 b.q1 a.q3 b.q3]
__device__ __forceinline__ void swap16(float &a, float &b)
{
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// in: reg r of quarter g holds element 4*g + r;  out: reg r of quarter g holds element 4*r + g
__device__ __forceinline__ void transpose4(float &r0, float &r1, float &r2, float &r3)
{
    swap32(r0, r2);  // r0 = [0 4 2 6], r2 = [8 12 10 14]
    swap32(r1, r3);  // r1 = [1 5 3 7], r3 = [9 13 11 15]
    swap16(r0, r1);  // r0 = [0 1 2 3], r1 = [4 5 6 7]
    swap16(r2, r3);  // r2 = [8 9 10 11], r3 = [12 13 14 15]
}

// 4x4 transpose of a register quadruple across the four lanes of a QUAD (lanes 4Q .. 4Q + 3): after the call r[k] of lane
// q (= lane & 3) holds what r[q] of lane 4Q + k held.  Two butterfly stages (lane bit <-> register bit), each one
// v_cndmask to pick what is sent, one DPP quad_perm move, two v_cndmask to place what arrives: 16 vector instructions.
// Use: a D[co][pixel] accumulator gives every lane (= pixel) four 16-byte pieces of its own 128-byte output row; stored as
// they stand, one store instruction touches 32 rows with 32 bytes each.  Transposed over the quad, store k of a quad's
// lanes carries the four pieces of ONE pixel (pixel 4Q + k): with both lane halves that is the whole 128-byte row per
// instruction and quad -- the store shape of a row-major epilogue, without leaving the registers.
__device__ __forceinline__ float dpp_quad_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose4(float &r0, float &r1, float &r2, float &r3, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const float s01 = dpp_quad_xor1(b0 ? r0 : r1), s23 = dpp_quad_xor1(b0 ? r2 : r3);
    r0 = b0 ? s01 : r0; r1 = b0 ? r1 : s01;
    r2 = b0 ? s23 : r2; r3 = b0 ? r3 : s23;
    const float s02 = dpp_quad_xor2(b1 ? r0 : r2), s13 = dpp_quad_xor2(b1 ? r1 : r3);
    r0 = b1 ? s02 : r0; r2 = b1 ? r2 : s02;
    r1 = b1 ? s13 : r1; r3 = b1 ? r3 : s13;
}
__device__ __forceinline__ void quad_transpose4(float4 &a, float4 &b, float4 &c, float4 &d, int lane)
{
    quad_transpose4(a.x, b.x, c.x, d.x, lane);
    quad_transpose4(a.y, b.y, c.y, d.y, lane);
    quad_transpose4(a.z, b.z, c.z, d.z, lane);
    quad_transpose4(a.w, b.w, c.w, d.w, lane);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float prelu1(float v, float a) { return v >= 0.0f ? v : a * v; }

// Raw buffer access (uniform 128-bit resource in SGPRs + 32-bit lane offset + uniform offset): no
// per-lane 64-bit address registers, and the hardware range check (offset >= bytes) returns 0 / drops
// the store.  Weight fragments are addressed as  lane part (VGPR) + step part (SGPR / immediate).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

__device__ __forceinline__ float bload(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ float4 bload4(rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Phase timestamps of one wave (measurement aid; compiled in only with -DSSAL_PHASE_TRACE, see
// tools/phase_trace.py): marks are s_memtime shader-clock reads kept in SGPRs, flushed by lane 0 at the
// end of the kernel: 16 x u64 per wave = t[0..11], realtime(100 MHz) first / last mark, HW_ID, XCC_ID.
struct PhaseTrace {
#ifdef SSAL_PHASE_TRACE
    unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long r0 = 0, r1 = 0;
#endif
    __device__ __forceinline__ void mark(int k)
    {
#ifdef SSAL_PHASE_TRACE
        t[k] = __builtin_amdgcn_s_memtime();
        if (k == 0) r0 = __builtin_amdgcn_s_memrealtime();
#endif
    }
    __device__ __forceinline__ void flush(unsigned long long *buf, int lane, int wave)
    {
#ifdef SSAL_PHASE_TRACE
        r1 = __builtin_amdgcn_s_memrealtime();
        if (buf && lane == 0) {
            unsigned long long *p = buf + ((long)blockIdx.x * 4 + wave) * 16;
            for (int k = 0; k < 12; ++k) p[k] = t[k];
            p[12] = r0; p[13] = r1;
            p[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
            p[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        }
#endif
    }
};

// 32x32x2 family: register order in which pair-swapped accumulator registers deliver ascending
// channel pairs: 0,2,1,3, 4,6,5,7, ...  (swap bits 0 and 1 of the step index)
__device__ __host__ constexpr int ord(int s) { return (s & ~3) | ((s & 1) << 1) | ((s >> 1) & 1); }

}  // namespace ssal
