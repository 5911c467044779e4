// Functor-driven MFMA tile GEMM for gfx950.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]          (both operands presented "k-contiguous")
//
// Operands are produced by *loader functors* (arbitrary addressing: plain row-major,
// transposed, im2col gathers, pool-routed gradients ...) as fp32 and handed tile-by-tile to an
// *epilogue functor* (bias/ReLU/pool/optimizer fusions live there). The element type T of the LDS
// tiles is the precision policy of the product:
//   T = float  : operands stay fp32, v_mfma_f32_32x32x2_f32 - bit-for-bit an fp32 fma chain (the
//                precision of the reference, whose Trainer runs 32-bit:
//                fedstellar/learning/pytorch/lightninglearner.py:182-193). Default.
//   T = __bf16 : operands rounded to bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16 with fp32
//                accumulation (16x the matrix rate; the labelled fast mode).
//
// Geometry (64-lane waves):  block = WR x WC waves, block tile BM x BN, BK = 32 (default; 64 / 128 for
// weight-streaming products with a long reduction, where a 32-deep step is latency- not bandwidth-bound).
// Each wave owns (BM/WR) x (BN/WC) outputs = RM x RN tiles of 32x32.
// LDS rows hold BK elements + 16 B pad (bf16: 80 / 144 / 272 B, fp32: 144 / 272 / 528 B = an odd multiple
// of 16 B): every 16-lane ds_read_b128 group lands on 16 distinct 16-B slots of the 256-B bank row, so
// fragment reads are conflict-free. An fp32 fragment read hands lane (r, h) the four values k = 4h .. 4h+3 of
// an 8-deep chunk; MFMA e of the chunk multiplies element e of both operands, i.e. reduces the k pair
// (e, 4 + e) - the order inside the reduction is free as long as both operands agree on it.
// Staging is register-split (issue global loads for tile t+1, run the MFMAs of tile t,
// then write t+1 into the other LDS buffer): one barrier per K-step.
//
// Loader contract:
//   static constexpr bool ROW_CONTIG;   // true: consecutive lanes -> consecutive rows
//                                       // false: consecutive lanes -> consecutive k-chunks of a row
//   __device__ void load8(int row, int k, float (&v)[8]) const;  // row,k global; k % 8 == 0;
//                                       // must zero-fill anything out of range
// Epilogue contract:
//   __device__ void operator()(const f32x16& acc, int m0, int n0) const;
//       lane l holds column n0 + (l & 31), rows m0 + (reg&3) + 8*(reg>>2) + 4*(l>>5).
//   N4 epilogues: static constexpr int BATCH (1: store4(v, m, n) per row; > 1: struct Row + load4(Row&, m, n) for
//       BATCH rows at once, then finish4(Row&, v, m, n) per row).
#pragma once
#include "common.h"

constexpr int GEMM_BK = 32;

template <class T> struct is_f32 { static constexpr bool value = false; };
template <> struct is_f32<float> { static constexpr bool value = true; };
// elements of padding per LDS row (16 B)
template <class T> constexpr int lds_pad() { return 16 / (int)sizeof(T); }

__device__ __forceinline__ void store_chunk(__bf16* dst, const float (&v)[8]) {
    *reinterpret_cast<bf16x8*>(dst) = pack_bf16x8(v);
}
__device__ __forceinline__ void store_chunk(float* dst, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

__device__ __forceinline__ int mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// PERM4 (ROWS % 128 == 0; per 128-row block): LDS row rho holds global row 4*(rho & 31) + (rho >> 5). With four
// 32-column MFMA tiles per wave this puts output columns 4r..4r+3 into the four accumulators of
// lane r, so an epilogue can move float4 (16 B) per lane along the contiguous output dimension.
template <int ROWS, int NT, class Loader, bool PERM4 = false, int BK = GEMM_BK, class T = __bf16>
struct TileStager {
    static constexpr int KC = BK / 8;                          // chunks per row
    static constexpr int LDK = BK + lds_pad<T>();
    static constexpr int CHUNKS = ROWS * KC;                   // 8-element chunks per tile
    static constexpr int PER_THREAD = (CHUNKS + NT - 1) / NT;  // chunks each thread moves
    float v[PER_THREAD][8];

    __device__ __forceinline__ static void decode(int c, int& row, int& k8) {
        if (Loader::ROW_CONTIG) { row = c % ROWS; k8 = c / ROWS; }
        else                    { row = c / KC;   k8 = c % KC; }
    }
    __device__ __forceinline__ void load(const Loader& ld, int row0, int k0) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int c = threadIdx.x + i * NT;
            if (CHUNKS % NT == 0 || c < CHUNKS) {
                int row, k8; decode(c, row, k8);
                if (PERM4) row = (row & ~127) + 4 * (row & 31) + ((row & 127) >> 5);
                ld.load8(row0 + row, k0 + k8 * 8, v[i]);
            }
        }
    }
    __device__ __forceinline__ void store(T* lds) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            int c = threadIdx.x + i * NT;
            if (CHUNKS % NT == 0 || c < CHUNKS) {
                int row, k8; decode(c, row, k8);
                store_chunk(lds + row * LDK + k8 * 8, v[i]);
            }
        }
    }
};

// NBUF = 2: double-buffered staging (one barrier per k-step). NBUF = 1: a single buffer (two barriers
// per k-step) for products whose reduction is one or two k-steps deep, where the second buffer would
// only cost occupancy (weight gradients: k = batch = 32).
template <int BM, int BN, int NBUF = 2, int BK = GEMM_BK, class T = __bf16>
struct GemmSmem {
    T a[NBUF][BM * (BK + lds_pad<T>())];
    T b[NBUF][BN * (BK + lds_pad<T>())];
};

// Runs one BM x BN block tile over k in [k_begin, k_end) (k_begin % 32 == 0; loaders
// zero-fill past their own K extent; k_begin % BK == 0). All threads of the block must call it.
// N4 = true (needs 128-column wave tiles, BN = 128 * WC): B is staged with the PERM4 row permutation and the epilogue
// is called as  epi.store4(float4 v, int m, int n)  with v = C[m][n .. n+3], n % 4 == 0.
template <int BM, int BN, int WR, int WC, class AL, class BL, class EPI, bool N4 = false, int NBUF = 2,
          int BK = GEMM_BK, class T = __bf16>
__device__ __forceinline__ void gemm_block_tile(GemmSmem<BM, BN, NBUF, BK, T>& sm, const AL& al, const BL& bl,
                                                const EPI& epi, int m0, int n0, int k_begin,
                                                int k_end) {
    constexpr int NT = 64 * WR * WC;
    constexpr int LDK = BK + lds_pad<T>();
    constexpr int HK = is_f32<T>::value ? 4 : 8;     // elements per lane half of a fragment read
    constexpr int TM = BM / WR, TN = BN / WC;
    constexpr int RM = TM / 32, RN = TN / 32;
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of 32x32");

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wr = wave / WC, wc = wave % WC;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    TileStager<BM, NT, AL, false, BK, T> sa;
    TileStager<BN, NT, BL, N4, BK, T> sb;
    static_assert(!N4 || TN == 128, "N4 epilogue needs 128-column wave tiles");

    sa.load(al, m0, k_begin);
    sb.load(bl, n0, k_begin);
    sa.store(sm.a[0]);
    sb.store(sm.b[0]);
    __syncthreads();

    int cur = 0;
    for (int k = k_begin; k < k_end; k += BK) {
        const bool has_next = (k + BK) < k_end;
        if (has_next) {
            sa.load(al, m0, k + BK);
            sb.load(bl, n0, k + BK);
        }
        const T* as = sm.a[cur] + (wr * TM + r) * LDK + h * HK;
        const T* bs = sm.b[cur] + (wc * TN + r) * LDK + h * HK;
        if constexpr (is_f32<T>::value) {
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 af[RM], bfr[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDK + kk * 8);
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    bfr[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDK + kk * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int j = 0; j < RN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 af[RM], bfr[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(as + i * 32 * LDK + kk * 16);
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8*>(bs + j * 32 * LDK + kk * 16);
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (NBUF == 2) {
            if (has_next) {
                sa.store(sm.a[cur ^ 1]);
                sb.store(sm.b[cur ^ 1]);
            }
            __syncthreads();
            cur ^= 1;
        } else {
            if (has_next) {
                __syncthreads();          // every wave is done reading the only buffer
                sa.store(sm.a[0]);
                sb.store(sm.b[0]);
                __syncthreads();
            }
        }
    }

    if constexpr (N4) {
        if constexpr (EPI::BATCH > 1) {
            // read-modify-write epilogues (fused optimizers): the loads of EPI::BATCH rows are issued together, then
            // the rows are finished one by one - BATCH x the bytes in flight of the load-wait-store row loop
            constexpr int G = EPI::BATCH;
            static_assert(16 % G == 0, "batch must divide the 16 accumulator rows");
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int e0 = 0; e0 < 16; e0 += G) {
                    typename EPI::Row rows[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        epi.load4(rows[u], m0 + wr * TM + i * 32 + mfma_row(e0 + u, lane), n0 + wc * TN + 4 * r);
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        epi.finish4(rows[u], make_float4(acc[i][0][e0 + u], acc[i][1][e0 + u], acc[i][2][e0 + u],
                                                         acc[i][3][e0 + u]),
                                    m0 + wr * TM + i * 32 + mfma_row(e0 + u, lane), n0 + wc * TN + 4 * r);
                }
        } else {
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    epi.store4(make_float4(acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]),
                               m0 + wr * TM + i * 32 + mfma_row(e, lane), n0 + wc * TN + 4 * r);
        }
    } else {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
                epi(acc[i][j], m0 + wr * TM + i * 32, n0 + wc * TN + j * 32);
    }
}

// ---------------------------------------------------------------------------------------
// Plain loaders
// ---------------------------------------------------------------------------------------

// src[row * ld + k], k contiguous. `vec` enables the float4 fast path; the host launcher
// sets it only when ld % 4 == 0 and the base (incl. client stride) is 16-B aligned.
struct LoadRowMajorK {
    static constexpr bool ROW_CONTIG = false;
    const float* src; int rows, kdim, ld; bool vec;
    const int* idx = nullptr;      // optional gather: logical row r is src row idx[r] (mini-batch index list)
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        const int srow = (idx != nullptr && row < rows) ? idx[row] : row;
        if (vec && row < rows && k + 8 <= kdim) {
            const float4* p = reinterpret_cast<const float4*>(src + (size_t)srow * ld + k);
            float4 a = p[0], b = p[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (row < rows && k + j < kdim) ? src[(size_t)srow * ld + k + j] : 0.f;
        }
    }
};

// src[k * ld + row], row contiguous (the operand is stored transposed w.r.t. its GEMM role).
struct LoadColMajorK {
    static constexpr bool ROW_CONTIG = true;
    const float* src; int rows, kdim, ld;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (row < rows && k + j < kdim) ? src[(size_t)(k + j) * ld + row] : 0.f;
    }
};
