// Client-batched convolution stages of the MNIST / FEMNIST CNNs for gfx950.
//
// Stage 1 (C_in = 1, 5x5 or 7x7, "same" padding, + ReLU + 2x2 max-pool) is a direct LDS-tiled
// VALU kernel: K = 25/49 is too thin for the matrix cores and the layer is <4 % of the FLOPs.
// Its weight gradient and all of stage 2 (C_in = 32 -> C_out = 64, 5x5 or 3x3, + ReLU + pool)
// run on the matrix cores through the functor GEMM of gemm_core.h: the loaders below perform the
// im2col gather / pool-gradient routing on the fly, so neither the im2col matrix, nor the
// pre-pool activations, nor the (75 % zero) pre-pool gradients are ever written to HBM.
//
// Pool/ReLU bookkeeping: every pooled output stores a 1-byte code q in {0,1,2,3} = position
// (dy*2+dx) of the window maximum, or 4 when the maximum was <= 0 (ReLU kills the gradient).
//
// Reference layers: fedstellar/learning/pytorch/mnist/models/cnn.py:134-142,151-154 and
// fedstellar/learning/pytorch/femnist/models/cnn.py:136-139,147-148.
#include "gemm_core.h"

namespace {

constexpr int IMG = 28;         // input H = W
constexpr int P1 = 14;          // after pool 1
constexpr int P2 = 7;           // after pool 2
constexpr int CELLS1 = P1 * P1; // 196
constexpr int CELLS2 = P2 * P2; // 49

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// Stage 1 forward: x[K][n][28*28] (gathered through sidx) -> a1[K][B][F][14][14], code1
// ---------------------------------------------------------------------------------------
struct Conv1Args {
    const float* data; long data_cs;   // [K][n_samples][784]
    const int* sidx;   long sidx_cs;   // [K][B] sample indices (or null: identity)
    const float* w;    long w_cs;      // [K][F][KS*KS]
    const float* bias; long b_cs;      // [K][F]
    float* a1;         long a_cs;      // [K][B][F][196]
    uint8_t* code;     long c_cs;      // [K][B][F][196]
    int B, F;
};

template <int KS>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1Args a) {
    constexpr int PAD = KS / 2;
    constexpr int TW = IMG + 2 * PAD;       // padded tile width
    constexpr int TWS = TW + 1;             // +1 float: odd stride, column reads spread over banks
    constexpr int PW = KS + 1;              // input patch feeding one 2x2 window
    __shared__ float img[TW * TWS];
    __shared__ float wsm[32 * KS * KS];
    __shared__ float bsm[32];

    const int c = blockIdx.y, b = blockIdx.x;
    const int sample = a.sidx ? a.sidx[c * a.sidx_cs + b] : b;
    const float* src = a.data + c * a.data_cs + (size_t)sample * (IMG * IMG);
    for (int i = threadIdx.x; i < TW * TWS; i += 256) {
        const int y = i / TWS - PAD, x = i % TWS - PAD;
        img[i] = (y >= 0 && y < IMG && x >= 0 && x < IMG) ? src[y * IMG + x] : 0.f;
    }
    const float* w = a.w + c * a.w_cs;
    for (int i = threadIdx.x; i < a.F * KS * KS; i += 256) wsm[i] = w[i];
    if (threadIdx.x < a.F) bsm[threadIdx.x] = a.bias[c * a.b_cs + threadIdx.x];
    __syncthreads();

    const int t = threadIdx.x;
    if (t >= CELLS1) return;
    const int py = t / P1, px = t % P1;
    float patch[PW][PW];
#pragma unroll
    for (int i = 0; i < PW; ++i)
#pragma unroll
        for (int j = 0; j < PW; ++j) patch[i][j] = img[(2 * py + i) * TWS + 2 * px + j];

    float* out = a.a1 + c * a.a_cs + (size_t)b * a.F * CELLS1 + t;
    uint8_t* code = a.code + c * a.c_cs + (size_t)b * a.F * CELLS1 + t;
    for (int f = 0; f < a.F; ++f) {
        float s00 = bsm[f], s01 = s00, s10 = s00, s11 = s00;
        const float* wf = wsm + f * KS * KS;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const float wv = wf[ky * KS + kx];
                s00 = fmaf(wv, patch[ky][kx], s00);
                s01 = fmaf(wv, patch[ky][kx + 1], s01);
                s10 = fmaf(wv, patch[ky + 1][kx], s10);
                s11 = fmaf(wv, patch[ky + 1][kx + 1], s11);
            }
        float best = s00; int q = 0;
        if (s01 > best) { best = s01; q = 1; }
        if (s10 > best) { best = s10; q = 2; }
        if (s11 > best) { best = s11; q = 3; }
        if (!(best > 0.f)) { best = 0.f; q = 4; }
        out[(size_t)f * CELLS1] = best;
        code[(size_t)f * CELLS1] = (uint8_t)q;
    }
}

// ---------------------------------------------------------------------------------------
// Stage 1 weight gradient on the matrix cores (split-K over the batch, fp32 atomics):
//   dW1[f][ky,kx] = sum_p G[p][f] * X[p][ky,kx],   p = ((b*196 + cell)*4 + q)  (pre-pool pixels)
//   G routes da1 to the window's arg-max pixel; column KS*KS of X is all ones -> bias grad.
// ---------------------------------------------------------------------------------------
struct LoadRoutedGrad1 {   // A[m = f][k = p]
    static constexpr bool ROW_CONTIG = false;
    const float* g; const uint8_t* code; int F, B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row >= F) return;
        const int L = k >> 2;   // pooled linear index (even), L and L+1 are in the same image
        if (L >= B * CELLS1) return;
        const int b = L / CELLS1, cell = L % CELLS1;
        const size_t o = ((size_t)b * F + row) * CELLS1 + cell;
        const int q0 = code[o], q1 = code[o + 1];
        const float g0 = g[o], g1 = g[o + 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // select form: a runtime index would spill v[] to scratch
            v[q] = (q0 == q) ? g0 : 0.f;
            v[4 + q] = (q1 == q) ? g1 : 0.f;
        }
    }
};

template <int KS>
struct LoadPatchPixels1 {  // B[n = ky*KS+kx | KS*KS -> ones][k = p]
    static constexpr bool ROW_CONTIG = false;
    const float* data; const int* sidx; int B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        constexpr int PAD = KS / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row > KS * KS) return;
        const int L = k >> 2;
        if (L >= B * CELLS1) return;
        if (row == KS * KS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 1.f;
            return;
        }
        const int b = L / CELLS1, cell = L % CELLS1;   // cell even -> cell, cell+1 share a row
        const int ky = row / KS, kx = row % KS;
        const int sample = sidx ? sidx[b] : b;
        const float* img = data + (size_t)sample * (IMG * IMG);
        const int py = cell / P1, px = cell % P1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cc = j >> 2, q = j & 3;
            const int y = 2 * py + (q >> 1) + ky - PAD;
            const int x = 2 * (px + cc) + (q & 1) + kx - PAD;
            if (y >= 0 && y < IMG && x >= 0 && x < IMG) v[j] = img[y * IMG + x];
        }
    }
};

struct EpiAtomicWgrad {
    float* dw; float* db; int M, N;   // dw[M][N]; column N -> db[m]
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n > N) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < M) {
                if (n < N) atomicAdd(dw + (size_t)m * N + n, acc[e]);
                else atomicAdd(db + m, acc[e]);
            }
        }
    }
};

struct Conv1WgradArgs {
    const float* g;    long g_cs;      // da1 [K][B][F][196]
    const uint8_t* code; long c_cs;
    const float* data; long data_cs;
    const int* sidx;   long sidx_cs;
    float* dw;         long dw_cs;     // [K][F][KS*KS]  (must be zeroed)
    float* db;         long db_cs;     // [K][F]         (must be zeroed)
    int B, F, k_per_block;
};

template <int KS, int BN>
__global__ __launch_bounds__(64) void conv1_wgrad_kernel(Conv1WgradArgs a) {
    __shared__ GemmSmem<32, BN> sm;
    const int c = blockIdx.z;
    LoadRoutedGrad1 al{a.g + c * a.g_cs, a.code + c * a.c_cs, a.F, a.B};
    LoadPatchPixels1<KS> bl{a.data + c * a.data_cs, a.sidx ? a.sidx + c * a.sidx_cs : nullptr, a.B};
    EpiAtomicWgrad epi{a.dw + c * a.dw_cs, a.db + c * a.db_cs, a.F, KS * KS};
    const int kd = a.B * CELLS1 * 4;
    const int k0 = blockIdx.x * a.k_per_block;
    const int k1 = min(kd, k0 + a.k_per_block);
    if (k0 < k1) gemm_block_tile<32, BN, 1, 1>(sm, al, bl, epi, 0, 0, k0, k1);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
P2PFL_API int p2pfl_conv1_fwd(const float* data, long data_cs, const int* sidx, long sidx_cs,
                              const float* w, long w_cs, const float* bias, long b_cs, float* a1,
                              long a_cs, uint8_t* code, long c_cs, int K, int B, int F, int KS,
                              hipStream_t stream) {
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    Conv1Args a{data, data_cs, sidx, sidx_cs, w, w_cs, bias, b_cs, a1, a_cs, code, c_cs, B, F};
    dim3 grid(B, K);
    if (KS == 5) hipLaunchKernelGGL(conv1_fwd_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL(conv1_fwd_kernel<7>, grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_conv1_wgrad(const float* g, long g_cs, const uint8_t* code, long c_cs,
                                const float* data, long data_cs, const int* sidx, long sidx_cs,
                                float* dw, long dw_cs, float* db, long db_cs, int K, int B, int F,
                                int KS, hipStream_t stream) {
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    // dw/db are accumulated with atomics: the caller zeroes them first (p2pfl_zero_rows)
    const int kd = B * CELLS1 * 4;
    int splits = cdiv(256 * 8, K);           // aim for >= ~2k single-wave blocks in flight
    if (splits < 1) splits = 1;
    int kpb = cdiv(cdiv(kd, splits), 32) * 32;
    if (kpb < 256) kpb = 256;
    splits = cdiv(kd, kpb);
    Conv1WgradArgs a{g, g_cs, code, c_cs, data, data_cs, sidx, sidx_cs, dw, dw_cs, db, db_cs, B, F, kpb};
    dim3 grid(splits, 1, K);
    if (KS == 5) hipLaunchKernelGGL((conv1_wgrad_kernel<5, 32>), grid, dim3(64), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL((conv1_wgrad_kernel<7, 64>), grid, dim3(64), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}
