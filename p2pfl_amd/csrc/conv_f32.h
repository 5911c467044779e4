// Launchers of the fp32 (v_mfma_f32_32x32x2_f32) convolution kernels; the C ABI entry points in conv1.hip /
// conv2.hip dispatch here when their `prec` argument is PREC_F32. All return a hipError_t as int (0 == ok).
#pragma once
#include "common.h"
#include "opt_epi.h"

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
                    int flags, const P2pflOpt* opt, hipStream_t stream);

// stage 1 (conv1_f32.hip)
int conv1_fwd_f32(const float* data, long data_cs, const int* sidx, long sidx_cs, const float* w, long w_cs,
                  const float* bias, long b_cs, float* a1, long a_cs, uint8_t* code, long c_cs, int K, int B, int F,
                  int KS, hipStream_t stream);
int conv1_wgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* data, long data_cs,
                    const int* sidx, long sidx_cs, float* dw, long dw_cs, float* db, long db_cs, float* scratch,
                    long scratch_floats, int K, int B, int F, int KS, const P2pflOpt* opt, hipStream_t stream);
int conv1_wgrad_f32_splits(int K, int B);
// Fixed-order sum of the per-wave partial tiles of a stage-1 weight gradient (conv1.hip) + gradient store or optimizer
// step. ``tap8``: column n of a tile is tap (n >> 3, n & 7) with the bias at n = 63 (bf16 kernel), else tap n with the
// bias at n = KS*KS (fp32 kernel).
int conv1_wgrad_finish(const float* part, int nparts, int njn, int tap8, float* dw, long dw_cs, float* db, long db_cs,
                       int K, int F, int KS, const P2pflOpt* opt, hipStream_t stream);
