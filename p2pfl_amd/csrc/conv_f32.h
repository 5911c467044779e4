// Launchers of the fp32 (v_mfma_f32_32x32x2_f32) convolution kernels; the C ABI entry points in conv1.hip /
// conv2.hip dispatch here when their `prec` argument is PREC_F32. All return a hipError_t as int (0 == ok).
#pragma once
#include "common.h"

// flags of the stage-2 launchers: small-footprint variants (<= 67 KB LDS, <= 128 registers, 256 threads) that fit on
// a CU beside the two resident workgroups of the fused wgrad+Adam kernel - for the two-stream training step
constexpr int CONV_CORESIDENT = 1;

// stage 2 (conv2_f32.hip). wp: per client 2 * KS*KS * 64 * 32 floats: forward copy [t][ci][co], then the
// mirrored data-gradient copy [t'][co][ci].
int conv2_prep_f32(const float* w, long w_cs, float* wp, long wp_cs, int K, int KS, hipStream_t stream);
int conv2_fwd_f32(const float* a1, long a1_cs, const float* wp, long wp_cs, const float* bias, long b_cs, float* a2,
                  long a2_cs, uint8_t* code, long c_cs, int K, int B, int KS, int flags, hipStream_t stream);
int conv2_dgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* wp, long wp_cs,
                    float* da1, long da1_cs, int K, int B, int KS, int flags, hipStream_t stream);
long conv2_wgrad_f32_scratch(int K, int B, int KS);
int conv2_wgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* a1, long a1_cs, float* dw,
                    long dw_cs, float* db, long db_cs, float* scratch, long scratch_floats, int K, int B, int KS,
                    int flags, hipStream_t stream);

// stage 1 (conv1_f32.hip)
int conv1_fwd_f32(const float* data, long data_cs, const int* sidx, long sidx_cs, const float* w, long w_cs,
                  const float* bias, long b_cs, float* a1, long a_cs, uint8_t* code, long c_cs, int K, int B, int F,
                  int KS, hipStream_t stream);
int conv1_wgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* data, long data_cs,
                    const int* sidx, long sidx_cs, float* dw, long dw_cs, float* db, long db_cs, int K, int B, int F,
                    int KS, hipStream_t stream);
