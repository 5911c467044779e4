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

// ---------------------------------------------------------------------------------------
// Stage 2 forward: implicit GEMM,  M = B*196 (pool-ordered m = (b*49+cell)*4 + q), N = C_out,
// K = C_in*KS*KS. The four rows of a pool window sit in four consecutive accumulator
// registers of one lane, so bias + ReLU + max-pool + arg-max code happen in registers.
// ---------------------------------------------------------------------------------------
template <int CIN, int KS>
struct LoadIm2col2 {       // A[m][k = c*KS*KS + ky*KS + kx]
    static constexpr bool ROW_CONTIG = true;
    const float* a1; int B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        constexpr int PAD = KS / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        const int L = row >> 2, q = row & 3;
        if (L >= B * CELLS2 || k >= CIN * KS * KS) return;
        const int b = L / CELLS2, cell = L % CELLS2;
        const int y = 2 * (cell / P2) + (q >> 1) - PAD;
        const int x = 2 * (cell % P2) + (q & 1) - PAD;
        const float* src = a1 + (size_t)b * CIN * CELLS1;
        int ci = k / (KS * KS);
        int rem = k % (KS * KS);
        int ky = rem / KS, kx = rem % KS;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int yy = y + ky, xx = x + kx;
            if (ci < CIN && yy >= 0 && yy < P1 && xx >= 0 && xx < P1)
                v[j] = src[ci * CELLS1 + yy * P1 + xx];
            if (++kx == KS) { kx = 0; if (++ky == KS) { ky = 0; ++ci; } }
        }
    }
};

struct EpiBiasReluPool {
    float* a2; uint8_t* code; const float* bias; int B, COUT;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n >= COUT) return;
        const float bv = bias[n];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int L = (m0 >> 2) + 2 * g + (lane >> 5);   // (m0 + 8g + 4h) / 4
            if (L >= B * CELLS2) continue;
            float best = acc[4 * g]; int q = 0;
            if (acc[4 * g + 1] > best) { best = acc[4 * g + 1]; q = 1; }
            if (acc[4 * g + 2] > best) { best = acc[4 * g + 2]; q = 2; }
            if (acc[4 * g + 3] > best) { best = acc[4 * g + 3]; q = 3; }
            best += bv;
            if (!(best > 0.f)) { best = 0.f; q = 4; }
            const int b = L / CELLS2, cell = L % CELLS2;
            const size_t o = ((size_t)b * COUT + n) * CELLS2 + cell;
            a2[o] = best;
            code[o] = (uint8_t)q;
        }
    }
};

struct Conv2Args {
    const float* a1;   long a1_cs;     // [K][B][CIN][196]
    const float* w;    long w_cs;      // [K][COUT][CIN*KS*KS]
    const float* bias; long b_cs;      // [K][COUT]
    float* a2;         long a2_cs;     // [K][B][COUT][49]
    uint8_t* code;     long c_cs;
    int B, COUT, wvec;
};

template <int CIN, int KS>
__global__ __launch_bounds__(256) void conv2_fwd_kernel(Conv2Args a) {
    __shared__ GemmSmem<128, 64> sm;
    const int c = blockIdx.z;
    LoadIm2col2<CIN, KS> al{a.a1 + c * a.a1_cs, a.B};
    LoadRowMajorK bl{a.w + c * a.w_cs, a.COUT, CIN * KS * KS, CIN * KS * KS, (bool)a.wvec};
    EpiBiasReluPool epi{a.a2 + c * a.a2_cs, a.code + c * a.c_cs, a.bias + c * a.b_cs, a.B, a.COUT};
    gemm_block_tile<128, 64, 4, 1>(sm, al, bl, epi, blockIdx.y * 128, blockIdx.x * 64, 0,
                                   CIN * KS * KS);
}

// ---------------------------------------------------------------------------------------
// Stage 2 weight gradient:  dW2[n][k] = sum_p dOut[p][n] * im2col(a1)[p][k]
//   M = C_out, N = C_in*KS*KS (+1 ones column -> bias), K = B*196 pool-ordered pixels.
// ---------------------------------------------------------------------------------------
struct LoadRoutedGrad2 {   // A[m = n_out][k = p]
    static constexpr bool ROW_CONTIG = false;
    const float* g; const uint8_t* code; int COUT, B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row >= COUT) return;
        const int L0 = k >> 2;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int L = L0 + cc;
            if (L < B * CELLS2) {
                const int b = L / CELLS2, cell = L % CELLS2;
                const size_t o = ((size_t)b * COUT + row) * CELLS2 + cell;
                const int qc = code[o];
                const float gv = g[o];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * cc + q] = (qc == q) ? gv : 0.f;
            }
        }
    }
};

template <int CIN, int KS>
struct LoadIm2col2T {      // B[n = k_w (c,ky,kx) | CIN*KS*KS -> ones][k = p]
    static constexpr bool ROW_CONTIG = false;
    const float* a1; int B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        constexpr int PAD = KS / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row > CIN * KS * KS) return;
        const int L0 = k >> 2;
        const int ci = row / (KS * KS), rem = row % (KS * KS);
        const int ky = rem / KS, kx = rem % KS;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int L = L0 + cc;
            if (L >= B * CELLS2) continue;
            if (row == CIN * KS * KS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * cc + q] = 1.f;
                continue;
            }
            const int b = L / CELLS2, cell = L % CELLS2;
            const float* src = a1 + ((size_t)b * CIN + ci) * CELLS1;
            const int y0 = 2 * (cell / P2) + ky - PAD, x0 = 2 * (cell % P2) + kx - PAD;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
                if (yy >= 0 && yy < P1 && xx >= 0 && xx < P1) v[4 * cc + q] = src[yy * P1 + xx];
            }
        }
    }
};

struct EpiWgradStore2 {
    float* dw; float* db; int M, N;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n > N) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < M) {
                if (n < N) dw[(size_t)m * N + n] = acc[e];
                else db[m] = acc[e];
            }
        }
    }
};

struct Conv2WgradArgs {
    const float* g;    long g_cs;      // da2 [K][B][COUT][49]
    const uint8_t* code; long c_cs;
    const float* a1;   long a1_cs;
    float* dw;         long dw_cs;
    float* db;         long db_cs;
    int B, COUT;
};

template <int CIN, int KS>
__global__ __launch_bounds__(256) void conv2_wgrad_kernel(Conv2WgradArgs a) {
    __shared__ GemmSmem<64, 128> sm;
    const int c = blockIdx.z;
    LoadRoutedGrad2 al{a.g + c * a.g_cs, a.code + c * a.c_cs, a.COUT, a.B};
    LoadIm2col2T<CIN, KS> bl{a.a1 + c * a.a1_cs, a.B};
    EpiWgradStore2 epi{a.dw + c * a.dw_cs, a.db + c * a.db_cs, a.COUT, CIN * KS * KS};
    gemm_block_tile<64, 128, 2, 2>(sm, al, bl, epi, blockIdx.y * 64, blockIdx.x * 128, 0,
                                   a.B * CELLS2 * 4);
}

// ---------------------------------------------------------------------------------------
// Stage 2 data gradient:  da1[m][c] = sum_{n,ky,kx} dOut[b][n][y-ky+P][x-kx+P] * W[n][c][ky][kx]
//   M = B*196 (natural m = b*196 + y*14 + x), N = C_in, K = C_out*KS*KS, k = n*KS*KS + ky*KS + kx.
// ---------------------------------------------------------------------------------------
template <int KS>
struct LoadShiftedGrad2 {  // A[m][k]
    static constexpr bool ROW_CONTIG = true;
    const float* g; const uint8_t* code; int COUT, B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        constexpr int PAD = KS / 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row >= B * CELLS1 || k >= COUT * KS * KS) return;
        const int b = row / CELLS1, yx = row % CELLS1;
        const int y = yx / P1 + PAD, x = yx % P1 + PAD;
        int n = k / (KS * KS), rem = k % (KS * KS);
        int ky = rem / KS, kx = rem % KS;
        const size_t base = (size_t)b * COUT * CELLS2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int yy = y - ky, xx = x - kx;
            if (n < COUT && yy >= 0 && yy < P1 && xx >= 0 && xx < P1) {
                const size_t o = base + (size_t)n * CELLS2 + (yy >> 1) * P2 + (xx >> 1);
                if (code[o] == ((yy & 1) * 2 + (xx & 1))) v[j] = g[o];
            }
            if (++kx == KS) { kx = 0; if (++ky == KS) { ky = 0; ++n; } }
        }
    }
};

template <int CIN, int KS>
struct LoadWeightsT2 {     // B[n = c][k = n_out*KS*KS + ky*KS + kx] = W[n_out][c][ky][kx]
    static constexpr bool ROW_CONTIG = false;
    const float* w; int COUT;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        int n = k / (KS * KS), rem = k % (KS * KS);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (row < CIN && n < COUT) ? w[((size_t)n * CIN + row) * (KS * KS) + rem] : 0.f;
            if (++rem == KS * KS) { rem = 0; ++n; }
        }
    }
};

struct EpiStoreDa1 {
    float* da1; int B, CIN;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n >= CIN) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < B * CELLS1) {
                const int b = m / CELLS1, yx = m % CELLS1;
                da1[((size_t)b * CIN + n) * CELLS1 + yx] = acc[e];
            }
        }
    }
};

struct Conv2DgradArgs {
    const float* g;    long g_cs;
    const uint8_t* code; long c_cs;
    const float* w;    long w_cs;
    float* da1;        long da1_cs;
    int B, COUT;
};

template <int CIN, int KS>
__global__ __launch_bounds__(256) void conv2_dgrad_kernel(Conv2DgradArgs a) {
    __shared__ GemmSmem<128, 32> sm;
    const int c = blockIdx.z;
    LoadShiftedGrad2<KS> al{a.g + c * a.g_cs, a.code + c * a.c_cs, a.COUT, a.B};
    LoadWeightsT2<CIN, KS> bl{a.w + c * a.w_cs, a.COUT};
    EpiStoreDa1 epi{a.da1 + c * a.da1_cs, a.B, CIN};
    gemm_block_tile<128, 32, 4, 1>(sm, al, bl, epi, blockIdx.y * 128, 0, 0, a.COUT * KS * KS);
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

P2PFL_API int p2pfl_conv2_fwd(const float* a1, long a1_cs, const float* w, long w_cs,
                              const float* bias, long b_cs, float* a2, long a2_cs, uint8_t* code,
                              long c_cs, int K, int B, int CIN, int COUT, int KS,
                              hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN != 32 || COUT <= 0) return -1;
    const int kw = CIN * KS * KS;
    const int wvec = !((kw & 3) || (w_cs & 3) || ((uintptr_t)w & 15));
    Conv2Args a{a1, a1_cs, w, w_cs, bias, b_cs, a2, a2_cs, code, c_cs, B, COUT, wvec};
    dim3 grid(cdiv(COUT, 64), cdiv(B * CELLS1, 128), K);
    if (KS == 5) hipLaunchKernelGGL((conv2_fwd_kernel<32, 5>), grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL((conv2_fwd_kernel<32, 3>), grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_conv2_wgrad(const float* g, long g_cs, const uint8_t* code, long c_cs,
                                const float* a1, long a1_cs, float* dw, long dw_cs, float* db,
                                long db_cs, int K, int B, int CIN, int COUT, int KS,
                                hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN != 32 || COUT <= 0) return -1;
    Conv2WgradArgs a{g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, B, COUT};
    dim3 grid(cdiv(CIN * KS * KS + 1, 128), cdiv(COUT, 64), K);
    if (KS == 5) hipLaunchKernelGGL((conv2_wgrad_kernel<32, 5>), grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL((conv2_wgrad_kernel<32, 3>), grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_conv2_dgrad(const float* g, long g_cs, const uint8_t* code, long c_cs,
                                const float* w, long w_cs, float* da1, long da1_cs, int K, int B,
                                int CIN, int COUT, int KS, hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN != 32 || COUT <= 0) return -1;
    Conv2DgradArgs a{g, g_cs, code, c_cs, w, w_cs, da1, da1_cs, B, COUT};
    dim3 grid(1, cdiv(B * CELLS1, 128), K);
    if (KS == 5) hipLaunchKernelGGL((conv2_dgrad_kernel<32, 5>), grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL((conv2_dgrad_kernel<32, 3>), grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}
