// Stage-1 convolution (C_in = 1; 5x5 or 7x7, "same" padding, + ReLU + 2x2 max-pool) of the MNIST / FEMNIST
// CNNs in full fp32 on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains), client-batched.
// The reference-precision siblings of the bf16 kernels in conv1.hip; same decomposition, but an fp32 MFMA takes
// ONE float of each operand per lane (lane l: A[row l & 31][k = l >> 5], B[k = l >> 5][col l & 31]), so every
// fragment is a ds_read_b32 at (per-lane base + immediate) and the k order is free to choose:
//
//   forward   C_q[f][window] = sum_k W[f][k] X_q[window][k]   one wave = one image in a private LDS copy, 7 tiles of
//             32 pool windows, one accumulator per window position q. k pairs are (ky, kx) | (ky + 1, kx): the lane
//             half selects the filter row, folded into the base address; filter taps live in registers.
//   weight    dW[f][n] = sum_{q, window} G_q[f][window] X_q[window][n]   n = tap (ky * KS + kx), n = KS*KS is the
//   gradient  all-ones column (bias gradient); k pairs are windows (px, px + 7) of one window row, so a lane half
//             owns 7 consecutive windows and nothing is padded; the pooled gradient is routed to the arg-max
//             pixel by comparing the code with q while the operand is built. Image de-interleaved by column
//             parity so that the 7 windows are 7 consecutive floats. Blocks stream the images of a client through
//             double-buffered LDS, the four waves split the 14 window rows, fp32 atomics of the per-wave partials.
//
// Reference layers: fedstellar/learning/pytorch/mnist/models/cnn.py:134-138,151-152 (5x5),
// fedstellar/learning/pytorch/femnist/models/cnn.py:136,147 (7x7).
#include <cstdlib>
#include "conv_f32.h"

namespace {

constexpr int IMG = 28;
constexpr int P1 = 14;
constexpr int CELLS1 = 196;

__device__ __forceinline__ int mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

struct Conv1Args {
    const float* data; long data_cs;   // [K][n_samples][784]
    const int* sidx;   long sidx_cs;   // [K][B] sample indices (or null: identity)
    const float* w;    long w_cs;      // [K][F][KS*KS]
    const float* bias; long b_cs;      // [K][F]
    float* a1;         long a_cs;      // [K][B][F][196]
    uint8_t* code;     long c_cs;      // [K][B][F][196]
    int B, F, ipw;
};

template <int KS>
__global__ __launch_bounds__(256) void conv1_fwd_f32_kernel(Conv1Args a) {
    constexpr int PAD = KS / 2;
    constexpr int NKK = (KS + 1) / 2;           // filter-row pairs: 3 / 4
    constexpr int ROWS = IMG + 2 * NKK;         // last pixel row 27 + last tap row KS - 1 (+ the phantom tap T of an odd T)
    constexpr int LW = 36;                      // last pixel column 27 + kx 6 = 33
    constexpr int NPX = IMG * IMG / 4;          // float4 per image
    constexpr int OT_LD = 40, CT_LD = 48;       // output patch rows: 32 windows + pad (bank spread / 16-B rows)
    // one private image + output patch per wave: no workgroup barrier anywhere in this kernel
    __shared__ __attribute__((aligned(16))) float img[4][ROWS * LW];
    __shared__ __attribute__((aligned(16))) float otile[4][32 * OT_LD];
    __shared__ __attribute__((aligned(16))) uint8_t ctile[4][32 * CT_LD];

    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    float* my = img[wave];
    float* ot = otile[wave];
    uint8_t* ct = ctile[wave];
    const int ipw = a.ipw;                                   // images per wave (weights / bias loaded once)
    int b = (blockIdx.x * ipw) * 4 + wave;
    if (b >= a.B) return;
    const float* data = a.data + c * a.data_cs;
    const int* sidx = a.sidx ? a.sidx + c * a.sidx_cs : nullptr;
    auto image = [&](int bi) {
        return reinterpret_cast<const f32x4*>(data + (size_t)(sidx ? sidx[bi] : bi) * (IMG * IMG));
    };
    f32x4 px[4];
    {
        const f32x4* src = image(b);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < NPX) px[i] = src[lane + 64 * i];
    }
    for (int i = lane; i < ROWS * LW; i += 64) my[i] = 0.f;   // zero border, once
    // A operand: the taps of filter r in PAIRS of consecutive taps (2 i, 2 i + 1) - lane half h holds tap 2 i + h - i.e.
    // ceil(T / 2) = 13 / 25 MFMAs per accumulator; pairing filter ROWS (ky, ky + 1) costs 3 x 5 = 15 / 4 x 7 = 28 with a
    // phantom row. The image offset of tap 2 i + 1 is that of tap 2 i plus 1, or plus LW - (KS - 1) where the pair wraps
    // to the next filter row: two per-lane base pointers, every read stays base + immediate.
    constexpr int T = KS * KS, NPAIR = (T + 1) / 2;
    float wf[NPAIR];
    {
        const float* w = a.w + c * a.w_cs;
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) wf[i] = (r < a.F && 2 * i + h < T) ? w[r * T + 2 * i + h] : 0.f;
    }
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int f = mfma_row(e, lane);
        bv[e] = f < a.F ? a.bias[c * a.b_cs + f] : 0.f;
    }
    for (int it = 0; it < ipw && b < a.B; ++it, b += 4) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // previous image consumed / border zeroed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = lane + 64 * i;
            if (o < NPX) {
                const int y = o / (IMG / 4), x = 4 * (o - y * (IMG / 4));
                float* dst = my + (y + PAD) * LW + x + PAD;
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = px[i][j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (it + 1 < ipw && b + 4 < a.B) {                       // next image in flight during the MFMAs
            const f32x4* src = image(b + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane + 64 * i < NPX) px[i] = src[lane + 64 * i];
        }
        float* out = a.a1 + c * a.a_cs + (size_t)b * a.F * CELLS1;
        uint8_t* code = a.code + c * a.c_cs + (size_t)b * a.F * CELLS1;
        for (int tile = 0; tile < (CELLS1 + 31) / 32; ++tile) {
            const int W = tile * 32 + r;
            const int Wc = min(W, CELLS1 - 1);
            const int py = Wc / P1, pxx = Wc - py * P1;
            float best[16]; int qi[16];
#pragma unroll
            for (int half = 0; half < 2; ++half) {               // window rows dy = 0, 1: two accumulators each
                f32x16 acc[2];
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
                const float* p = my + (2 * py + half) * LW + 2 * pxx;
                const float* p1 = p + h;                          // pairs inside a filter row
                const float* p2 = p + h * (LW - (KS - 1));        // pairs that wrap to the next row
#pragma unroll
                for (int i = 0; i < NPAIR; ++i) {
                    const int t0 = 2 * i, ky = t0 / KS, kx = t0 % KS;
                    const float* q = (kx == KS - 1) ? p2 : p1;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i], q[ky * LW + kx + d], acc[d], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (half == 0) { best[e] = acc[0][e]; qi[e] = 0; }
                    else if (acc[0][e] > best[e]) { best[e] = acc[0][e]; qi[e] = 2; }
                    if (acc[1][e] > best[e]) { best[e] = acc[1][e]; qi[e] = 2 * half + 1; }
                }
            }
            // bias + ReLU, then the tile (32 windows x 32 filters) goes through a per-wave LDS patch so that the
            // stores are 16 bytes per lane in address order
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = best[e] + bv[e]; int q = qi[e];
                if (!(v > 0.f)) { v = 0.f; q = 4; }
                const int f = mfma_row(e, lane);
                ot[f * OT_LD + r] = v;
                ct[f * CT_LD + r] = (uint8_t)q;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane, f = idx >> 3, j = idx & 7;
                if (f < a.F && tile * 32 + 4 * j < CELLS1)
                    *reinterpret_cast<f32x4*>(out + (size_t)f * CELLS1 + tile * 32 + 4 * j) =
                        *reinterpret_cast<const f32x4*>(ot + f * OT_LD + 4 * j);
            }
            {
                const int f = lane >> 1, hf = lane & 1, w0 = tile * 32 + 16 * hf;
                uint8_t* dst = code + (size_t)f * CELLS1 + w0;
                if (f < a.F && w0 + 16 <= CELLS1)
                    *reinterpret_cast<u32x4_a4*>(dst) = *reinterpret_cast<const u32x4_a4*>(ct + f * CT_LD + 16 * hf);
                else if (f < a.F && w0 < CELLS1)          // last tile: windows 192..195
                    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(ct + f * CT_LD + 16 * hf);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
}

// ---------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------
struct Conv1WgradArgs {
    const float* g;    long g_cs;      // da1 [K][B][F][196]
    const uint8_t* code; long c_cs;
    const float* data; long data_cs;
    const int* sidx;   long sidx_cs;
    float* dw;         long dw_cs;     // [K][F][KS*KS]  (written by the finishing pass, unless it applies the optimizer)
    float* db;         long db_cs;     // [K][F]
    int B, F;
    float* part;                       // per-wave partial tiles [K][gridDim.x * 4][NJN][16][64] (raw accumulator layout)
};

constexpr int WG_ROWS = 36;     // last row 2*13 + 1 + tap row 6 = 33
constexpr int WG_HW = 24;       // half-row: last index 13 + (1 + 6 >> 1) = 16
constexpr int WG_IMG = WG_ROWS * WG_HW;
constexpr int WG_GS = CELLS1 + 1;      // odd filter stride of the gradient rows: lanes = filters hit 32 banks
constexpr int WG_CS = CELLS1 + 4;      // code rows (bytes): 50 dwords - 2-way at worst, words stay aligned

struct Conv1WgradSmemF {
    float img[2][WG_IMG];       // [parity][row][i] = padded pixel (row, 2 i + parity)
    float g[32 * WG_GS];        // pooled gradient [f][196]
    uint8_t code[32 * WG_CS];   // arg-max codes   [f][196]
};

template <int KS>
struct Conv1WgradStageF {
    static constexpr int PAD = KS / 2;
    static constexpr int NG = (32 * CELLS1 / 4 + 255) / 256;   // float4 of the gradient per thread: 7
    static constexpr int NP = (2 * WG_IMG + 255) / 256;        // staged pixels per thread: 7
    f32x4 g[NG]; uint32_t c[NG]; float px[NP];

    __device__ __forceinline__ void load(const Conv1WgradArgs& a, int cl, int b) {
        const int n4 = a.F * CELLS1 / 4;
        const f32x4* gsrc = reinterpret_cast<const f32x4*>(a.g + cl * a.g_cs + (size_t)b * a.F * CELLS1);
        const uint32_t* csrc = reinterpret_cast<const uint32_t*>(a.code + cl * a.c_cs + (size_t)b * a.F * CELLS1);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int o = threadIdx.x + i * 256;
            if (o < n4) { g[i] = gsrc[o]; c[i] = csrc[o]; }
        }
        const int sample = a.sidx ? a.sidx[cl * a.sidx_cs + b] : b;
        const float* src = a.data + cl * a.data_cs + (size_t)sample * (IMG * IMG);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int o2 = threadIdx.x + i * 256;
            const int par = o2 / WG_IMG, o = o2 - par * WG_IMG;
            const int y = o / WG_HW - PAD, x = 2 * (o % WG_HW) + par - PAD;
            px[i] = (o2 < 2 * WG_IMG && y >= 0 && y < IMG && x >= 0 && x < IMG) ? src[y * IMG + x] : 0.f;
        }
    }
    __device__ __forceinline__ void store(const Conv1WgradArgs& a, Conv1WgradSmemF& sm) const {
        const int n4 = a.F * CELLS1 / 4;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int o = threadIdx.x + i * 256;
            if (o < n4) {
                const int f = o / (CELLS1 / 4), j = o - f * (CELLS1 / 4);     // 49 float4 per filter row
                float* gd = sm.g + f * WG_GS + 4 * j;
#pragma unroll
                for (int e = 0; e < 4; ++e) gd[e] = g[i][e];
                *reinterpret_cast<uint32_t*>(sm.code + f * WG_CS + 4 * j) = c[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int o2 = threadIdx.x + i * 256;
            if (o2 < 2 * WG_IMG) (&sm.img[0][0])[o2] = px[i];
        }
    }
};

template <int KS>
__global__ __launch_bounds__(256) void conv1_wgrad_f32_kernel(Conv1WgradArgs a) {
    constexpr int T = KS * KS;
    constexpr int NJN = (T + 1 + 31) / 32;       // column tiles of (taps + bias): 1 / 2
    __shared__ __attribute__((aligned(16))) Conv1WgradSmemF smem[2];

    const int c = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    // rows of filters >= F stay "no gradient" for the whole kernel
    for (int i = threadIdx.x; i < 2 * 32 * WG_GS; i += 256) smem[i / (32 * WG_GS)].g[i % (32 * WG_GS)] = 0.f;
    for (int i = threadIdx.x; i < 2 * 32 * WG_CS; i += 256) smem[i / (32 * WG_CS)].code[i % (32 * WG_CS)] = 4;
    __syncthreads();
    f32x16 acc[NJN];
#pragma unroll
    for (int j = 0; j < NJN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // B operand roles: column n = jn * 32 + r is tap (ky, kx) = (n / KS, n % KS), n == T the ones column
    int ky[NJN], kx[NJN]; bool is_tap[NJN], is_one[NJN];
#pragma unroll
    for (int jn = 0; jn < NJN; ++jn) {
        const int n = jn * 32 + r;
        is_tap[jn] = n < T; is_one[jn] = n == T;
        ky[jn] = is_tap[jn] ? n / KS : 0;
        kx[jn] = is_tap[jn] ? n % KS : 0;
    }

    Conv1WgradStageF<KS> st;
    int b = blockIdx.x;
    if (b < a.B) { st.load(a, c, b); st.store(a, smem[0]); }
    __syncthreads();
    int cur = 0;
    for (; b < a.B; b += gridDim.x) {
        const bool has_next = b + (int)gridDim.x < a.B;
        if (has_next) st.load(a, c, b + gridDim.x);          // global loads fly while this image is consumed
        const Conv1WgradSmemF& sm = smem[cur];
        // wave q takes pool corner q of ALL 14 window rows: 14 x 7 x NJN MFMAs per wave and sample, the same for every wave
        // (rows dealt to the waves - 4 / 4 / 3 / 3 - made the sample as long as 16 corner rows on the heavy waves)
        const int q = wave;
#pragma unroll 2
        for (int py = 0; py < P1; ++py) {
            // pooled gradient + codes of the 7 windows px = 7h .. 7h+6 of filter r
            const float* gr = sm.g + r * WG_GS + py * P1 + 7 * h;
            const uint8_t* cr = sm.code + r * WG_CS + py * P1 + 7 * h;
            float gv[7]; int cv[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) { gv[j] = gr[j]; cv[j] = cr[j]; }
            const float* bp[NJN];
#pragma unroll
            for (int jn = 0; jn < NJN; ++jn) {
                // pixel of window px for tap (ky, kx): (2py + dy + ky, 2px + dx + kx)
                const int v = (q & 1) + kx[jn];
                bp[jn] = sm.img[v & 1] + (2 * py + (q >> 1) + ky[jn]) * WG_HW + 7 * h + (v >> 1);
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float av = cv[j] == q ? gv[j] : 0.f;
#pragma unroll
                for (int jn = 0; jn < NJN; ++jn) {
                    const float bvv = is_tap[jn] ? bp[jn][j] : (is_one[jn] ? 1.f : 0.f);
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc[jn], 0, 0, 0);
                }
            }
        }
        if (has_next) st.store(a, smem[cur ^ 1]);
        __syncthreads();
        cur ^= 1;
    }
    // acc[jn][e] = partial dW[f = mfma_row(e)][column n = jn*32 + r] of this wave's window rows: stored raw (coalesced,
    // no atomics); conv1_wgrad_finish_kernel adds the partials of a client up in a fixed order - bitwise reproducible
    float* pw = a.part + (((size_t)c * gridDim.x + blockIdx.x) * 4 + wave) * (NJN * 1024);
#pragma unroll
    for (int jn = 0; jn < NJN; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(jn * 16 + e) * 64 + lane] = acc[jn][e];
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

int conv1_fwd_f32(const float* data, long data_cs, const int* sidx, long sidx_cs, const float* w, long w_cs,
                  const float* bias, long b_cs, float* a1, long a_cs, uint8_t* code, long c_cs, int K, int B, int F,
                  int KS, hipStream_t stream) {
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    if (((uintptr_t)data & 15) || (data_cs & 3)) return -2;      // float4 loads of whole 784-pixel images
    if (((uintptr_t)a1 & 15) || (a_cs & 3) || ((uintptr_t)code & 3) || (c_cs & 3)) return -2;   // 16-byte output pieces
    int ipw = (int)(((long)K * B / 4) / 2048);
    if (ipw < 1) ipw = 1;
    if (ipw > 8) ipw = 8;
    Conv1Args a{data, data_cs, sidx, sidx_cs, w, w_cs, bias, b_cs, a1, a_cs, code, c_cs, B, F, ipw};
    dim3 grid(cdiv(B, 4 * ipw), K);
    if (KS == 5) hipLaunchKernelGGL(conv1_fwd_f32_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL(conv1_fwd_f32_kernel<7>, grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

int conv1_wgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* data, long data_cs,
                    const int* sidx, long sidx_cs, float* dw, long dw_cs, float* db, long db_cs, float* scratch,
                    long scratch_floats, int K, int B, int F, int KS, const P2pflOpt* opt, hipStream_t stream) {
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    if (((uintptr_t)g & 15) || (g_cs & 3) || ((uintptr_t)code & 3) || (c_cs & 3)) return -2;
    const int splits = conv1_wgrad_f32_splits(K, B);
    const int njn = (KS * KS + 1 + 31) / 32;
    if (scratch == nullptr || scratch_floats < (long)K * splits * 4 * njn * 1024) return -5;
    Conv1WgradArgs a{g, g_cs, code, c_cs, data, data_cs, sidx, sidx_cs, dw, dw_cs, db, db_cs, B, F, scratch};
    dim3 grid(splits, 1, K);
    if (KS == 5) hipLaunchKernelGGL(conv1_wgrad_f32_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL(conv1_wgrad_f32_kernel<7>, grid, dim3(256), 0, stream, a);
    else return -3;
    return conv1_wgrad_finish(scratch, splits * 4, njn, /*tap8=*/0, dw, dw_cs, db, db_cs, K, F, KS, opt, stream);
}

// blocks per client - matrix-pipe-bound: about two blocks per CU over the launch, and a block keeps >= 2 images to amortise
// its prologue (10 clients: 16 blocks per client 34 / 25 us for 7x7 / 5x5, 32: 35 / 27; cnn10-fully 317.2 -> 314.8 ms).
// Alone the kernel would like to stay within the 512 resident blocks (50 clients: 10 per client 78 us, 11: 97 us;
// tools/probe/c1w_splits.sh, P2PFL_C1W_SPLITS forces a count) but inside the step it runs beside the stage-2 weight
// gradient of the side stream and the finer split wins there: femnist50-ring 5: 60.4 ms, 8: 56.4, 10: 58.7 against 57.8
// for 11 on the same box, 11: 55.2, 12: 55.2, 16: 56.2, 32: 56.6 - the rule stays ceil(512 / K).
int conv1_wgrad_f32_splits(int K, int B) {
    static const int forced = [] { const char* e = getenv("P2PFL_C1W_SPLITS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced > B ? B : forced;
    int splits = cdiv(512, K);
    if (splits > (B + 1) / 2) splits = (B + 1) / 2;
    if (splits < 1) splits = 1;
    return splits;
}
