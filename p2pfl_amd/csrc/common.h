// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of p2pfl_amd.
// Wavefront = 64 lanes everywhere in this tree; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define P2PFL_API extern "C" __attribute__((visibility("default")))

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int WAVE = 64;

// Precision policy of the matrix products (argument `prec` of the launchers):
//   PREC_F32  - ("fp32-mfma") fp32 operands on v_mfma_f32_32x32x2_f32 everywhere: exact fp32 fma chains
//   PREC_BF16 - operands rounded to bf16, fp32 accumulation (16x the matrix-core rate)
//   PREC_F32_SPLIT - ("fp32", the default policy and the reference's precision) as PREC_F32, except that the stage-2 convolutions (the MFMA-bound kernels of the step) form their
//               fp32 products from three bf16 terms per operand, a = a0 + a1 + a2 EXACTLY (3 x 8 significand bits),
//               keeping the six products a_i * b_j with i + j <= 2 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//               every kept product is exact, the dropped ones are <= 2^-24 |a b| (below half an fp32 ulp of the
//               product), and six bf16 MFMAs cost 192 matrix-core cycles where the fp32 MFMAs of the same 16-deep
//               chunk cost 512. Everything else (storage, dense layers, stage 1) is identical to PREC_F32.
constexpr int PREC_BF16 = 0;
constexpr int PREC_F32 = 1;
constexpr int PREC_F32_SPLIT = 2;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 8 fp32 -> 8 bf16 (RNE; hipcc lowers the casts to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16x8 pack_bf16x8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    return r;
}

// 8 fp32 -> NS bf16 terms whose sum is the input (exactly for NS == 3; NS == 1 is the plain rounding): each term
// is the RNE rounding of what the previous terms left over, so terms shrink by >= 2^-8 per level.
template <int NS>
__device__ __forceinline__ void split_bf16x8(const float* v, bf16x8 (&out)[NS]) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = v[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        out[s] = pack_bf16x8(r);
        if (s + 1 < NS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] -= (float)out[s][i];
        }
    }
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Every launcher returns the hipError_t of its launch as an int (0 == ok) so the
// Python side can fail loudly instead of silently continuing.
#define P2PFL_RETURN_LAST_ERROR() return (int)hipGetLastError()
