// Stage-2 convolution (C_in = 32 -> C_out = 64, 5x5 or 3x3 "same", + ReLU + 2x2 max-pool) of the MNIST /
// FEMNIST CNNs in full fp32 on the gfx950 matrix cores: forward, data gradient, weight gradient.
//
// This is the reference-precision path (the reference trains 32-bit: fedstellar/learning/pytorch/
// lightninglearner.py:182-193): operands are never rounded, every product runs on
// v_mfma_f32_32x32x2_f32, whose result is bit-for-bit an fp32 fma chain.
//
// Design: "resident image, one dword per lane". The fp32 MFMA takes ONE float of A and ONE float of B per
// lane (lane l: A[row l & 31][k = l >> 5], B[k = l >> 5][col l & 31]), so operand fragments are plain
// ds_read_b32 and the only layout question is the bank pattern of 32 consecutive lanes:
//   * images are kept CHANNEL-MAJOR in LDS, [channel][padded pixel], with an ODD channel stride
//     (18 x 18 + 1 = 325 / 16 x 16 + 1 = 257 floats): when the lane index is the channel (weight gradient),
//     32 lanes hit 32 different banks; when the lane index is the pixel (forward / data gradient), lanes walk
//     along image rows. Channel-major is also the layout of the activations in memory (NCHW), so staging is
//     a straight copy of 196-float rows - no transposition anywhere, unlike the bf16 kernels (conv2.hip),
//     which need channel-last images for their 8-element fragments.
//   * the k index of a product is (channel pair | pixel pair); the lane half h picks the element of the
//     pair, folded into the per-lane base address, so every read in the unrolled loops is base + immediate.
//   * a filter tap is a constant address offset on the image (never an im2col matrix); per tap the
//     [k][n] weight slice (8 KB of a tap-major fp32 copy made once per step by conv2_prep_f32) is
//     double-buffered through LDS.
// The fp32 MFMA issues at 1/16 of the bf16 rate (64 cycles per 32x32x2), so unlike their bf16 siblings these
// kernels are matrix-pipe-bound: one ds_read_b32 (2 LDS cycles) per 64-cycle MFMA, and what matters is that
// every SIMD of every CU always has an MFMA to issue (work units dealt evenly to the waves).
//
// Reference layers: fedstellar/learning/pytorch/mnist/models/cnn.py:138-142,152-154 and
// fedstellar/learning/pytorch/femnist/models/cnn.py:137-139,147-148.
#include <type_traits>

#include "conv_f32.h"
#include "opt_epi.h"

#include <map>
#include <mutex>

namespace {

constexpr int P1 = 14, P2 = 7, CELLS1 = 196, CELLS2 = 49;
constexpr int CIN = 32, COUT = 64;

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int mfma_row_(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int KS>
struct GeoF {
    static constexpr int PAD = KS / 2;
    static constexpr int IW = P1 + 2 * PAD;          // padded image width (18 / 16)
    static constexpr int T = KS * KS;
    static constexpr int CS = (IW * IW) | 1;         // odd channel stride in floats (325 / 257)
};

// block id -> (client, sub-block) with all sub-blocks of a client on one XCD (ids round-robin over 8 XCDs)
__device__ __forceinline__ bool xcd_map(int per_client, int K, int& c, int& sub) {
    const int i = blockIdx.x;
    const int xcd = i & 7, j = i >> 3;
    c = (j / per_client) * 8 + xcd;
    sub = j % per_client;
    return c < K;
}

// ---------------------------------------------------------------------------------------
// weight prep: W[n][c][t] -> Wf[t][c][n] (forward B operand, lane = n) and Wd[t'][n][c] (data-gradient
// B operand, lane = c, t' = mirrored tap)
// ---------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv2_prep_f32_kernel(const float* w, long w_cs, float* wp, long wp_cs) {
    constexpr int T = KS * KS;
    const int c = blockIdx.y;
    const float* src = w + c * w_cs;
    float* wf = wp + c * wp_cs;
    float* wd = wf + T * COUT * CIN;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < COUT * CIN * T; o += gridDim.x * 256) {
        {   // wf[(t * CIN + ci) * COUT + n]
            const int n = o % COUT, tc = o / COUT;
            const int ci = tc % CIN, t = tc / CIN;
            wf[o] = src[(n * CIN + ci) * T + t];
        }
        {   // wd[((T - 1 - t) * COUT + n) * CIN + ci]
            const int ci = o % CIN, tn = o / CIN;
            const int n = tn % COUT, t = T - 1 - tn / COUT;
            wd[o] = src[(n * CIN + ci) * T + t];
        }
    }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
struct FwdArgs {
    const float* a1;   long a1_cs;     // [K][B][32][196]
    const float* wp;   long wp_cs;     // prepared weights
    const float* bias; long b_cs;      // [K][64]
    float* a2;         long a2_cs;     // [K][B][64][49]
    uint8_t* code;     long c_cs;
    int K, B;
};

// 2 samples per block: M = 98 pool windows x 4 pixels (pool-ordered: the four rows of a window are four
// consecutive accumulator registers of one lane -> bias + ReLU + max-pool + arg-max code in registers) = 13 row
// tiles, x 2 column halves = 26 work units dealt 7 / 7 / 6 / 6 to the four waves (wave & 1 = its column half).
//
// S = 1 is the small-footprint variant for the two-stream step (flags & CONV_CORESIDENT): 58 KB of LDS and <= 128
// registers, so that a block fits on a CU BESIDE the two resident workgroups of the fused wgrad+Adam kernel (2 x 37 KB,
// 2 x 192 registers per SIMD) and its MFMAs fill the matrix pipe that kernel leaves idle while it waits for HBM.
// 7 row tiles x 2 halves = 14 units dealt 4 / 4 / 3 / 3: less even, which does not matter under the update.
template <int KS, int S>
__device__ __forceinline__ void conv2_fwd_f32_body(const FwdArgs& a) {
    using G = GeoF<KS>;
    constexpr int NWIN = S * CELLS2;           // 98 / 49
    constexpr int NFULL = NWIN / 8;            // full row tiles: 12 / 6
    constexpr int REM = NWIN % 8;              // left-over windows: 2 / 1 (on the 16-row instruction, see below)
    constexpr int MU = NFULL / 2;              // row tiles per wave: 6 / 3
    static_assert(NFULL % 2 == 0 && REM <= 4, "tiles split evenly over the two tile parities; left-over fits 16 rows");
    constexpr int XN = S * CIN * G::CS;
    __shared__ __attribute__((aligned(16))) float ximg[(XN + 3) / 4 * 4];
    __shared__ __attribute__((aligned(16))) float wbuf[2][CIN * COUT];

    int c, sub;
    if (!xcd_map(cdiv(a.B, S), a.K, c, sub)) return;
    const int b0 = sub * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    for (int i = tid; i < (XN + 3) / 4; i += 256)
        reinterpret_cast<f32x4*>(ximg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wf = a.wp + c * a.wp_cs;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        reinterpret_cast<f32x4*>(wbuf[0])[tid + 256 * u] = reinterpret_cast<const f32x4*>(wf)[tid + 256 * u];
    __syncthreads();
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.a1 + c * a.a1_cs + (size_t)b0 * CIN * CELLS1);
        const int nval = min(S, a.B - b0) * CIN * (CELLS1 / 4);
        for (int i = tid; i < nval; i += 256) {
            const f32x4 v = src[i];
            const int sc = i / (CELLS1 / 4), j = i - sc * (CELLS1 / 4);
            float* dst = ximg + sc * G::CS + G::PAD * G::IW + G::PAD;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 4 * j + e, y = p / P1, x = p - y * P1;
                dst[y * G::IW + x] = v[e];
            }
        }
    }
    __syncthreads();

    const int half = wave & 1, tile0 = wave >> 1;
    int abase[MU];
#pragma unroll
    for (int i = 0; i < MU; ++i) {
        const int W = (tile0 + 2 * i) * 8 + (r >> 2);
        const int q = r & 3, s = W / CELLS2, cell = W % CELLS2;
        const int y = 2 * (cell / P2) + (q >> 1), x = 2 * (cell % P2) + (q & 1);
        abase[i] = (s * CIN + h) * G::CS + y * G::IW + x;
    }
    // The REM windows past the last full tile run on v_mfma_f32_16x16x4_f32 (lane l: row l & 15 = window (l & 15) >> 2,
    // corner l & 3 - rows past the real windows repeat the first one -, k = l >> 4; B: column 16 wave + (l & 15)): a
    // 49-window sample used to be 7 row tiles issued as 8 (two waves recomputed a clamped tile), now 6 + this.
    int lbase, lwoff;
    {
        const int lw = (lane & 15) >> 2;
        const int W = NFULL * 8 + (lw < REM ? lw : 0);
        const int q = lane & 3, s = W / CELLS2, cell = W % CELLS2;
        const int y = 2 * (cell / P2) + (q >> 1), x = 2 * (cell % P2) + (q & 1);
        lbase = (s * CIN + (lane >> 4)) * G::CS + y * G::IW + x;
        lwoff = (lane >> 4) * COUT + 16 * wave + (lane & 15);
    }
    f32x16 acc[MU];
    f32x4 accl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MU; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    int ky = 0, kx = 0;
#pragma unroll 1
    for (int t = 0; t < G::T; ++t) {
        f32x4 wnext[2];
        if (t + 1 < G::T) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                wnext[u] = reinterpret_cast<const f32x4*>(wf + (t + 1) * COUT * CIN)[tid + 256 * u];
        }
        const float* wb = wbuf[t & 1] + h * COUT + half * 32 + r;
        const int toff = ky * G::IW + kx;
        const float* ap[MU];
#pragma unroll
        for (int i = 0; i < MU; ++i) ap[i] = ximg + abase[i] + toff;
#pragma unroll
        for (int s = 0; s < CIN / 2; ++s) {
            const float bv = wb[2 * s * COUT];
#pragma unroll
            for (int i = 0; i < MU; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[i][2 * s * G::CS], bv, acc[i], 0, 0, 0);
        }
        {
            const float* al = ximg + lbase + toff;
            const float* bl = wbuf[t & 1] + lwoff;
#pragma unroll
            for (int s = 0; s < CIN / 4; ++s)
                accl = __builtin_amdgcn_mfma_f32_16x16x4f32(al[4 * s * G::CS], bl[4 * s * COUT], accl, 0, 0, 0);
        }
        if (t + 1 < G::T) {
#pragma unroll
            for (int u = 0; u < 2; ++u) reinterpret_cast<f32x4*>(wbuf[(t + 1) & 1])[tid + 256 * u] = wnext[u];
        }
        __syncthreads();
        if (++kx == KS) { kx = 0; ++ky; }
    }

    // ---- epilogue: bias + ReLU + 2x2 max-pool + arg-max code from the accumulators; the block's outputs (S samples
    // x [64][49], one contiguous piece of a2 and of the codes) are assembled in the now idle image buffer and leave
    // as 16-byte pieces per lane in address order.
    static_assert(XN * 4 >= S * COUT * CELLS2 * 5, "output staging must fit into the image buffer");
    float* out_s = ximg;
    uint8_t* code_s = reinterpret_cast<uint8_t*>(ximg + S * COUT * CELLS2);
    const int n = half * 32 + r;
    const float bv = a.bias[c * a.b_cs + n];
    auto pool = [&](float v0, float v1, float v2, float v3, float bias_n, int nn, int W) {
        float best = v0; int q = 0;
        if (v1 > best) { best = v1; q = 1; }
        if (v2 > best) { best = v2; q = 2; }
        if (v3 > best) { best = v3; q = 3; }
        best += bias_n;
        if (!(best > 0.f)) { best = 0.f; q = 4; }
        const int o = ((W / CELLS2) * COUT + nn) * CELLS2 + W % CELLS2;     // == the global layout of the block
        out_s[o] = best;
        code_s[o] = (uint8_t)q;
    };
#pragma unroll
    for (int i = 0; i < MU; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            pool(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], bv, n, (tile0 + 2 * i) * 8 + 2 * g + h);
    if ((lane >> 4) < REM) {     // 16 x 16 accumulator: lane = column, registers = rows 4 (lane >> 4) .. + 3 = one window
        const int nn = 16 * wave + (lane & 15);
        pool(accl[0], accl[1], accl[2], accl[3], a.bias[c * a.b_cs + nn], nn, NFULL * 8 + (lane >> 4));
    }
    __syncthreads();
    const int nvalid = min(S, a.B - b0) * COUT * CELLS2;
    float* a2 = a.a2 + c * a.a2_cs + (size_t)b0 * COUT * CELLS2;
    uint8_t* code = a.code + c * a.c_cs + (size_t)b0 * COUT * CELLS2;
    for (int i = tid; i < nvalid / 4; i += 256)
        reinterpret_cast<f32x4*>(a2)[i] = reinterpret_cast<const f32x4*>(out_s)[i];
    for (int i = tid; i < nvalid / 16; i += 256)
        reinterpret_cast<f32x4*>(code)[i] = reinterpret_cast<const f32x4*>(code_s)[i];
}

// ---------------------------------------------------------------------------------------
// data gradient: image = pooled gradient routed to the arg-max pixel while staging (only the 25 % of the
// pre-pool pixels that won their window are ever written; the rest of the zeroed image stays zero),
// M = natural pixels of one sample (7 row tiles), N = C_in, k = C_out pairs x mirrored taps.
// ---------------------------------------------------------------------------------------
struct DgradArgs {
    const float* g;    long g_cs;      // da2 [K][B][64][49]
    const uint8_t* code; long c_cs;
    const float* wp;   long wp_cs;
    float* da1;        long da1_cs;    // [K][B][32][196]
    int K, B;
};

// CH = channels of the gradient image resident at a time: 64 (one pass, 100 KB of LDS) or 32 (two passes over the
// taps, 50 KB and <= 128 registers: the variant that fits beside the fused update kernel, see the forward kernel).
// 196 pixels are six full 32-row tiles and four pixels (the first form issued 8 tiles: two per wave). The twelve (tile,
// channel half of the pass) units are dealt three to a wave - wave w reduces channels (w & 1) CH / 2 .. + CH / 2 - 1 over
// tiles 3 (w >> 1) + {0, 1, 2}; the two waves of a tile triple add their accumulators through LDS at the end, in a fixed
// order - and the four left-over pixels run on v_mfma_f32_16x16x4_f32: wave w = (16-column tile w & 1, channel half w >> 1).
template <int KS, int CH>
__device__ __forceinline__ void conv2_dgrad_f32_body(const DgradArgs& a) {
    using G = GeoF<KS>;
    constexpr int GN = CH * G::CS;
    constexpr int WU = CH * CIN / 4 / 256;      // float4 of a weight slice per thread: 2 / 1
    __shared__ __attribute__((aligned(16))) float gimg[(GN + 3) / 4 * 4];
    __shared__ __attribute__((aligned(16))) float wbuf[2][CH * CIN];

    int c, b;
    if (!xcd_map(a.B, a.K, c, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int kh = wave & 1, t0 = 3 * (wave >> 1);

    int abase[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = (t0 + i) * 32 + r;
        abase[i] = (h + kh * (CH / 2)) * G::CS + (p / P1) * G::IW + (p % P1);
    }
    // left-over pixels: lane l -> row l & 15 (pixel 192 + (l & 3), rows >= 4 repeat it), k = l >> 4; B column 16 (wave & 1)
    // + (l & 15); this wave's channels (wave >> 1) CH / 2 .. + CH / 2 - 1
    const int lp = 192 + (lane & 3);
    const int lbase = ((wave >> 1) * (CH / 2) + (lane >> 4)) * G::CS + (lp / P1) * G::IW + (lp % P1);
    const int lwoff = ((wave >> 1) * (CH / 2) + (lane >> 4)) * CIN + 16 * (wave & 1) + (lane & 15);
    f32x16 acc[3];
    f32x4 accl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

#pragma unroll 1
    for (int pass = 0; pass < COUT / CH; ++pass) {
        // mirrored-tap weights [t'][co][ci]: rows co of this pass are one contiguous piece of a tap slice
        const float* wd = a.wp + c * a.wp_cs + G::T * COUT * CIN + pass * CH * CIN;
        if (pass > 0) __syncthreads();        // everyone is done with the previous image
        for (int i = tid; i < (GN + 3) / 4; i += 256)
            reinterpret_cast<f32x4*>(gimg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < WU; ++u)
            reinterpret_cast<f32x4*>(wbuf[0])[tid + 256 * u] = reinterpret_cast<const f32x4*>(wd)[tid + 256 * u];
        __syncthreads();
        {
            const float* gsrc = a.g + c * a.g_cs + ((size_t)b * COUT + pass * CH) * CELLS2;
            const uint8_t* csrc = a.code + c * a.c_cs + ((size_t)b * COUT + pass * CH) * CELLS2;
            for (int task = tid; task < CH * CELLS2; task += 256) {
                const int co = task / CELLS2, cell = task - co * CELLS2;
                const float gv = gsrc[task];
                const int q = csrc[task];
                if (q < 4) {
                    const int Y = 2 * (cell / P2) + G::PAD + (q >> 1), X = 2 * (cell % P2) + G::PAD + (q & 1);
                    gimg[co * G::CS + Y * G::IW + X] = gv;
                }
            }
        }
        __syncthreads();

        int ky = 0, kx = 0;   // mirrored tap index: image offset (ky,kx), weights of tap (KS-1-ky, KS-1-kx)
#pragma unroll 1
        for (int t = 0; t < G::T; ++t) {
            f32x4 wnext[WU];
            if (t + 1 < G::T) {
#pragma unroll
                for (int u = 0; u < WU; ++u)
                    wnext[u] = reinterpret_cast<const f32x4*>(wd + (t + 1) * COUT * CIN)[tid + 256 * u];
            }
            const float* wb = wbuf[t & 1] + (h + kh * (CH / 2)) * CIN + r;
            const int toff = ky * G::IW + kx;
            const float* ap[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) ap[i] = gimg + abase[i] + toff;
#pragma unroll
            for (int s = 0; s < CH / 4; ++s) {
                const float bv = wb[2 * s * CIN];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[i][2 * s * G::CS], bv, acc[i], 0, 0, 0);
            }
            {
                const float* al = gimg + lbase + toff;
                const float* bl = wbuf[t & 1] + lwoff;
#pragma unroll
                for (int s = 0; s < CH / 8; ++s)
                    accl = __builtin_amdgcn_mfma_f32_16x16x4f32(al[4 * s * G::CS], bl[4 * s * CIN], accl, 0, 0, 0);
            }
            if (t + 1 < G::T) {
#pragma unroll
                for (int u = 0; u < WU; ++u) reinterpret_cast<f32x4*>(wbuf[(t + 1) & 1])[tid + 256 * u] = wnext[u];
            }
            __syncthreads();
            if (++kx == KS) { kx = 0; ++ky; }
        }
    }

    // ---- the odd waves hand their channel halves over (the loop's last barrier freed the image)
    float* red = gimg;                          // [2 triples][3 tiles][16][64] + left-over [4 waves][4][64]
    float* redl = red + 2 * 3 * 16 * 64;
    static_assert(GN >= 2 * 3 * 16 * 64 + 4 * 4 * 64, "the exchange must fit into the image buffer");
    if (kh) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(((wave >> 1) * 3 + i) * 16 + e) * 64 + lane] = acc[i][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) redl[(wave * 4 + e) * 64 + lane] = accl[e];
    __syncthreads();
    if (!kh) {     // lane = input channel, registers 4g..4g+3 = four consecutive pixels -> float4
        float* dst = a.da1 + c * a.da1_cs + ((size_t)b * CIN + r) * CELLS1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int p = (t0 + i) * 32 + 8 * g + 4 * h;
                const float* o = red + (((wave >> 1) * 3 + i) * 16 + 4 * g) * 64 + lane;
                *reinterpret_cast<float4*>(dst + p) = make_float4(acc[i][4 * g] + o[0], acc[i][4 * g + 1] + o[64],
                                                                  acc[i][4 * g + 2] + o[128], acc[i][4 * g + 3] + o[192]);
            }
    } else if (lane < 16) {
        // left-over pixels: wave 1 finishes column tile 0 (waves 0 + 2: the two channel halves), wave 3 column tile 1 (1 + 3)
        const int j = wave >> 1;
        const int ci = 16 * j + lane;
        const float* p0 = redl + (j * 4) * 64 + lane;
        const float* p1 = redl + ((2 + j) * 4) * 64 + lane;
        float* dst = a.da1 + c * a.da1_cs + ((size_t)b * CIN + ci) * CELLS1 + 192;
        *reinterpret_cast<float4*>(dst) = make_float4(p0[0] + p1[0], p0[64] + p1[64], p0[128] + p1[128], p0[192] + p1[192]);
    }
}

// ---------------------------------------------------------------------------------------
// weight gradient:  dW[n][c][t] = sum_{b,p} G[b][p][n] * X[b][p + t][c],   db[n] = sum G[b][p][n]
//
// GEMM per tap: M = C_out (2 row tiles), N = C_in (1 tile), k = pixel pairs (98 per sample). Both operands have
// the CHANNEL on the lane and the pixel along k - with channel-major images and odd channel strides every
// fragment read is a conflict-free ds_read_b32 (the bf16 kernel needs transposing reads here). A block owns
// (client, 1/NP of the taps, 1/NS of the batch) and walks its samples, so its accumulators are final sums: plain
// stores (or per-split partials summed by conv2_wgrad_reduce_f32_kernel), no atomics, bitwise reproducible.
// 8 waves = 4 tap groups x 2 row tiles; a wave shares its G value across its taps; the bias gradient is a running
// VALU sum of the same G values. The next sample's rows are fetched into registers while this one is multiplied.
// ---------------------------------------------------------------------------------------
struct WgradArgs {
    const float* g;    long g_cs;      // da2 [K][B][64][49]
    const uint8_t* code; long c_cs;
    const float* a1;   long a1_cs;     // [K][B][32][196]
    float* dw;         long dw_cs;     // [K][64][32*T]
    float* db;         long db_cs;     // [K][64]
    int K, B;
    float* part;       int NS;         // part != null: batch split, partials [K][NS][T*64*32 + 64]
    OptEpi opt;                        // optimizer applied by the reduce kernel (kind 0: it writes dw / db)
};

// MH = output channels per block: 64 (8 waves = 4 tap groups x 2 row tiles, 92 KB of LDS) or 32 (4 waves = 4 tap
// groups of ONE row tile, blockIdx -> (.., row tile), 67 KB, <= 128 registers with NP >= 2: the variant that fits
// beside the fused update kernel, see the forward kernel).
template <int KS, int NP, int MH>
__device__ __forceinline__ void conv2_wgrad_f32_body(const WgradArgs& a) {
    using G = GeoF<KS>;
    constexpr int T = G::T;
    constexpr int NTH = MH * 8;                          // threads: 512 / 256
    constexpr int NTAP = ((T + NP - 1) / NP + 3) / 4;    // most taps per wave
    constexpr int GCS = CELLS1 + 1;                      // 197: odd channel stride of the routed gradient
    constexpr int XN = CIN * G::CS, GN = MH * GCS;
    constexpr int NX4 = CIN * CELLS1 / 4;                // float4 of the activations of a sample: 1568
    constexpr int XU = (NX4 + NTH - 1) / NTH;            // per thread: 4 / 7
    constexpr int NG = MH * CELLS2;                      // pooled gradient values of a sample (and block): 3136 / 1568
    constexpr int GU = (NG + NTH - 1) / NTH;             // per thread: 7
    __shared__ __attribute__((aligned(16))) float ximg[(XN + 3) / 4 * 4];
    __shared__ __attribute__((aligned(16))) float gimg[(GN + 3) / 4 * 4];

    constexpr int NMT = COUT / MH;                       // row-tile blocks per (client, tap part, split): 1 / 2
    int c, sub;
    if (!xcd_map(NP * a.NS * NMT, a.K, c, sub)) return;
    const int mblk = sub % NMT, tp = (sub / NMT) % NP, split = sub / (NMT * NP);
    const int b_begin = split * a.B / a.NS, b_end = (split + 1) * a.B / a.NS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tg = MH == 64 ? wave >> 1 : wave;          // tap group
    const int mloc = MH == 64 ? wave & 1 : 0;            // row tile inside the block's gradient image
    const int mt = MH == 64 ? mloc : mblk;               // row tile of the product
    // taps of this block [p0, p0 + np) split over the four tap groups; (part 0, group 0) also owns the bias
    const int p0 = tp * T / NP, np = (tp + 1) * T / NP - p0;
    const int t_begin = p0 + tg * np / 4;
    const int ntap = p0 + (tg + 1) * np / 4 - t_begin;
    const bool do_bias = (tp == 0 && tg == 0);     // wave-uniform

    for (int i = tid; i < (XN + 3) / 4; i += NTH)
        reinterpret_cast<f32x4*>(ximg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < (GN + 3) / 4; i += NTH)
        reinterpret_cast<f32x4*>(gimg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* xp[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int t = min(t_begin + j, T - 1);
        xp[j] = ximg + r * G::CS + h + (t / KS) * G::IW + (t % KS);
    }
    const float* gp = gimg + (mloc * 32 + r) * GCS + h;

    // staging roles (global -> registers -> LDS)
    f32x4 xv[XU]; float gv[GU]; uint8_t qc[GU];
    auto prefetch = [&](int b) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.a1 + c * a.a1_cs + (size_t)b * CIN * CELLS1);
#pragma unroll
        for (int u = 0; u < XU; ++u)
            if (tid + NTH * u < NX4) xv[u] = src[tid + NTH * u];
        const float* gs = a.g + c * a.g_cs + ((size_t)b * COUT + (MH == 64 ? 0 : mblk * 32)) * CELLS2;
        const uint8_t* cs = a.code + c * a.c_cs + ((size_t)b * COUT + (MH == 64 ? 0 : mblk * 32)) * CELLS2;
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (tid + NTH * u < NG) { gv[u] = gs[tid + NTH * u]; qc[u] = cs[tid + NTH * u]; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = tid + NTH * u;
            if (i < NX4) {
                const int ch = i / (CELLS1 / 4), j = i - ch * (CELLS1 / 4);
                float* dst = ximg + ch * G::CS + G::PAD * G::IW + G::PAD;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int p = 4 * j + e, y = p / P1, x = p - y * P1;
                    dst[y * G::IW + x] = xv[u][e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = tid + NTH * u;
            if (i < NG) {
                const int co = i / CELLS2, cell = i - co * CELLS2;
                float* dst = gimg + co * GCS + 2 * (cell / P2) * P1 + 2 * (cell % P2);
                const int q = qc[u];
                dst[0] = q == 0 ? gv[u] : 0.f;
                dst[1] = q == 1 ? gv[u] : 0.f;
                dst[P1] = q == 2 ? gv[u] : 0.f;
                dst[P1 + 1] = q == 3 ? gv[u] : 0.f;
            }
        }
    };

    f32x16 acc[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float bsum = 0.f;           // bias gradient of channel mt*32 + r over this lane half's pixels (plain VALU adds)

    // MH == 64: the next sample's rows are fetched into registers while this one is multiplied. The small variant
    // has no registers for that (128 in all, 64 of them accumulators): it stages each sample in place, and the
    // exposed fetch latency is covered by the other workgroups of the CU (it runs under the fused update).
    constexpr bool PREFETCH = MH == 64;
    if (PREFETCH && b_begin < b_end) prefetch(b_begin);
    for (int item = b_begin; item < b_end; ++item) {
        __syncthreads();            // everyone finished reading the previous images (or zero-filling)
        if constexpr (PREFETCH) {
            commit();
        } else {        // rolled loops, a few loads in flight each: no register arrays, no hoisted address tables
            const f32x4* src = reinterpret_cast<const f32x4*>(a.a1 + c * a.a1_cs + (size_t)item * CIN * CELLS1);
#pragma unroll 2
            for (int i = tid; i < NX4; i += NTH) {
                const f32x4 v = src[i];
                const int ch = i / (CELLS1 / 4), j = i - ch * (CELLS1 / 4);
                float* dst = ximg + ch * G::CS + G::PAD * G::IW + G::PAD;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int p = 4 * j + e, y = p / P1, x = p - y * P1;
                    dst[y * G::IW + x] = v[e];
                }
            }
            const float* gs = a.g + c * a.g_cs + ((size_t)item * COUT + mblk * 32) * CELLS2;
            const uint8_t* cs = a.code + c * a.c_cs + ((size_t)item * COUT + mblk * 32) * CELLS2;
#pragma unroll 4
            for (int i = tid; i < NG; i += NTH) {
                const float g = gs[i];
                const int q = cs[i];
                const int co = i / CELLS2, cell = i - co * CELLS2;
                float* dst = gimg + co * GCS + 2 * (cell / P2) * P1 + 2 * (cell % P2);
                dst[0] = q == 0 ? g : 0.f;
                dst[1] = q == 1 ? g : 0.f;
                dst[P1] = q == 2 ? g : 0.f;
                dst[P1 + 1] = q == 3 ? g : 0.f;
            }
        }
        __syncthreads();
        if (PREFETCH && item + 1 < b_end) prefetch(item + 1);

        // the sample's MFMA loop with a compile-time tap count (wave-uniform choice): a group with one tap less used to
        // recompute its last tap - 16 MFMAs per k-step and block for the 12 / 13 taps of a 5x5 half
        auto mm = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;
#pragma unroll 1
            for (int y = 0; y < P1; ++y) {
#pragma unroll
                for (int xs = 0; xs < P1 / 2; ++xs) {
                    const float av = gp[y * P1 + 2 * xs];
                    bsum += av;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xp[j][y * G::IW + 2 * xs], acc[j], 0, 0, 0);
                }
            }
        };
        if (ntap == NTAP) mm(std::integral_constant<int, NTAP>{});
        else if (ntap == NTAP - 1) mm(std::integral_constant<int, NTAP - 1>{});
        else mm(std::integral_constant<int, (NTAP > 2 ? NTAP - 2 : 0)>{});
    }

    const int ci = r;
    bsum += __shfl_xor(bsum, 32, 64);      // both pixel parities
    if (a.part != nullptr) {
        // partial of this batch split, tap-major [t][n][ci], then the bias
        float* pw = a.part + ((size_t)c * a.NS + split) * (T * COUT * CIN + COUT);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            if (j < ntap) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pw[((t_begin + j) * COUT + mt * 32 + mfma_row_(e, lane)) * CIN + ci] = acc[j][e];
            }
        }
        if (do_bias && h == 0) pw[T * COUT * CIN + mt * 32 + r] = bsum;
        return;
    }
    float* dw = a.dw + c * a.dw_cs;
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        if (j < ntap) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = mt * 32 + mfma_row_(e, lane);
                dw[((size_t)n * CIN + ci) * T + t_begin + j] = acc[j][e];
            }
        }
    }
    if (do_bias && h == 0) a.db[c * a.db_cs + mt * 32 + r] = bsum;
}

// Kernel entry points. The *_small variants are capped at 128 registers (VGPR + AGPR): with the 2 x 192 of the two
// resident fused-update workgroups that is exactly one SIMD's file, i.e. a block of theirs fits beside them.
template <int KS>
__global__ __launch_bounds__(256) void conv2_fwd_f32_kernel(FwdArgs a) { conv2_fwd_f32_body<KS, 2>(a); }
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128)))
void conv2_fwd_f32_small_kernel(FwdArgs a) { conv2_fwd_f32_body<KS, 1>(a); }
template <int KS>
__global__ __launch_bounds__(256) void conv2_dgrad_f32_kernel(DgradArgs a) { conv2_dgrad_f32_body<KS, 64>(a); }
template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128)))
void conv2_dgrad_f32_small_kernel(DgradArgs a) { conv2_dgrad_f32_body<KS, 32>(a); }
template <int KS, int NP>
__global__ __launch_bounds__(512) void conv2_wgrad_f32_kernel(WgradArgs a) { conv2_wgrad_f32_body<KS, NP, 64>(a); }
template <int KS, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128)))
void conv2_wgrad_f32_small_kernel(WgradArgs a) { conv2_wgrad_f32_body<KS, NP, 32>(a); }

// Sum of the NS batch-split partials [t][n][ci] -> dW[n][ci][t], db[n]. One block per (client, n).
template <int KS>
__global__ __launch_bounds__(256) void conv2_wgrad_reduce_f32_kernel(WgradArgs a) {
    constexpr int T = KS * KS, NW = T * COUT * CIN, PW = NW + COUT;
    __shared__ float tr[CIN * T];
    const int c = blockIdx.y, n = blockIdx.x;
    const float* p = a.part + (size_t)c * a.NS * PW;
    for (int i = threadIdx.x; i < CIN * T; i += 256) {
        const int t = i / CIN, ci = i % CIN;
        float s = 0.f;
        for (int k = 0; k < a.NS; ++k) s += p[(size_t)k * PW + (t * COUT + n) * CIN + ci];
        tr[ci * T + t] = s;
    }
    const OptEpi::Scal sc = a.opt.kind ? a.opt.scal() : OptEpi::Scal{0.f, 0.f};
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < a.NS; ++k) s += p[(size_t)k * PW + NW + n];
        if (a.opt.kind) a.opt.bias(sc, c, n, s);
        else a.db[c * a.db_cs + n] = s;
    }
    __syncthreads();
    if (a.opt.kind) {      // the final gradient is in hand: update filter n in place, the gradient never reaches memory
        for (int i = threadIdx.x; i < CIN * T; i += 256) a.opt.weight(sc, c, (long)n * CIN * T + i, tr[i]);
        return;
    }
    float* dw = a.dw + c * a.dw_cs + (size_t)n * CIN * T;
    for (int i = threadIdx.x; i < CIN * T; i += 256) dw[i] = tr[i];
}

// Optional dynamic LDS padding of a small-variant block (P2PFL_CORES_LDS_KB, experiments): limits how many of them the
// dispatcher packs onto a CU beside the fused-update workgroups. Measured without effect - what slows the update
// under fp32 convolutions is the MFMA issue arbitration, not occupancy (profiles/r2_overlap_probe.txt).
size_t coresident_pad(const void* kernel) {
    static std::mutex mu;
    static std::map<const void*, size_t> pads;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pads.find(kernel);
    if (it != pads.end()) return it->second;
    hipFuncAttributes attr;
    size_t pad = 0;
    const char* e = getenv("P2PFL_CORES_LDS_KB");
    const size_t TARGET = (e ? (size_t)atoi(e) : 0) * 1024;      // measured 0 / 64 / 72 / 80 / 100 KB: no difference
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess && attr.sharedSizeBytes < TARGET)
        pad = TARGET - attr.sharedSizeBytes;
    pads[kernel] = pad;
    return pad;
}
#define LAUNCH_SMALL(kernel, grid, block, stream, args) \
    hipLaunchKernelGGL(kernel, grid, block, coresident_pad(reinterpret_cast<const void*>(&kernel)), stream, args)

// The small-footprint variants are the default everywhere: two or three of their blocks share a CU, so one block's
// staging / epilogue hides under another's MFMAs (femnist1000-star 99.8 -> 95.6 ms, femnist50-ring 99.3 -> 88.6 ms
// per round; MNIST-CNN shard alone: fwd 0.745 -> 0.717, dgrad 0.820 -> 0.736, wgrad 0.855 -> 0.790 ms).
// P2PFL_CONV_SMALL=0 selects the one-block-per-CU variants for A/B runs.
bool want_small(int flags) {
    static const char* e = getenv("P2PFL_CONV_SMALL");
    (void)flags;
    return e ? atoi(e) != 0 : true;
}

int num_cus() {
    static int n = 0;
    if (n <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

// Work split of the fp32 weight gradient: np tap parts x ns batch splits per client. The kernel is matrix-pipe-
// bound and runs one 8-wave block per CU, so what counts is how evenly the MFMAs are dealt: whole waves of
// blocks over the CUs, samples over the batch splits, taps over the four tap groups of a block. Tap parts
// multiply the staging work and batch splits the partial-sum traffic (small penalties).
//   5x5: 100 clients -> 1 x 5 (500 blocks = 1.95 waves), 10 -> 5 x 5;  3x3: 1000 -> 1 x 1, 50 -> 1 x 5
// small (co-resident variant): one row tile per block (twice the blocks), np >= 2 (register budget)
void wgrad_split_f32(int K, int B, int KS, int cus, bool small, int& np, int& ns) {
    // small: <= 4 accumulators per wave (128-register budget): >= 2 tap parts
    static const int parts_big[3] = {1, 2, 5}, parts_small5[3] = {2, 3, 5}, parts_small3[3] = {1, 2, 3};
    const int* parts = !small ? parts_big : (KS == 5 ? parts_small5 : parts_small3);
    const int T = KS * KS;
    double best = -1.0;
    np = parts[0]; ns = 1;
    for (int i = 0; i < 3; ++i) {
        const int p = parts[i];
        if (p > T) continue;
        const int ntap = ((T + p - 1) / p + 3) / 4;
        const double tapbal = (double)T / (p * 4 * ntap);
        for (int s = 1; s <= 8 && s <= B; ++s) {
            const long blocks = (long)K * p * s * (small ? 2 : 1);
            const double fill = (double)blocks / ((double)((blocks + cus - 1) / cus) * cus);
            const double bbal = (double)B / (s * ((B + s - 1) / s));
            const double e = tapbal * fill * bbal / (1.0 + 0.05 * (p - 1)) / (1.0 + 0.01 * (s - 1));
            if (e > best) { best = e; np = p; ns = s; }
        }
    }
}

template <int KS>
void launch_wgrad(const WgradArgs& a, int np, bool small, hipStream_t stream) {
    const int slots = cdiv(a.K, 8) * 8 * a.NS;
    if (small) {
        const dim3 block(256);
        if (np == 1 && KS == 3) LAUNCH_SMALL((conv2_wgrad_f32_small_kernel<3, 1>), dim3(slots * 2), block, stream, a);
        else if (np <= 2) LAUNCH_SMALL((conv2_wgrad_f32_small_kernel<KS, 2>), dim3(slots * 4), block, stream, a);
        else if (np == 3) LAUNCH_SMALL((conv2_wgrad_f32_small_kernel<KS, 3>), dim3(slots * 6), block, stream, a);
        else LAUNCH_SMALL((conv2_wgrad_f32_small_kernel<KS, 5>), dim3(slots * 10), block, stream, a);
    } else {
        const dim3 block(512);
        if (np == 1) hipLaunchKernelGGL((conv2_wgrad_f32_kernel<KS, 1>), dim3(slots), block, 0, stream, a);
        else if (np == 2) hipLaunchKernelGGL((conv2_wgrad_f32_kernel<KS, 2>), dim3(slots * 2), block, 0, stream, a);
        else hipLaunchKernelGGL((conv2_wgrad_f32_kernel<KS, 5>), dim3(slots * 5), block, 0, stream, a);
    }
    if (a.part != nullptr)
        hipLaunchKernelGGL(conv2_wgrad_reduce_f32_kernel<KS>, dim3(COUT, a.K), dim3(256), 0, stream, a);
}

}  // namespace

int conv2_prep_f32(const float* w, long w_cs, float* wp, long wp_cs, int K, int KS, hipStream_t stream) {
    if (K <= 0) return -1;
    if (wp_cs < 2L * KS * KS * COUT * CIN || (wp_cs & 3) || ((uintptr_t)wp & 15)) return -2;
    dim3 grid(cdiv(COUT * CIN * KS * KS, 256 * 4), K);
    if (KS == 5) hipLaunchKernelGGL(conv2_prep_f32_kernel<5>, grid, dim3(256), 0, stream, w, w_cs, wp, wp_cs);
    else if (KS == 3) hipLaunchKernelGGL(conv2_prep_f32_kernel<3>, grid, dim3(256), 0, stream, w, w_cs, wp, wp_cs);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

int conv2_fwd_f32(const float* a1, long a1_cs, const float* wp, long wp_cs, const float* bias, long b_cs, float* a2,
                  long a2_cs, uint8_t* code, long c_cs, int K, int B, int KS, int flags, hipStream_t stream) {
    if (K <= 0 || B <= 0) return -1;
    if ((wp_cs & 3) || ((uintptr_t)wp & 15) || ((uintptr_t)a1 & 15) || (a1_cs & 3)) return -2;
    if (((uintptr_t)a2 & 15) || (a2_cs & 3) || ((uintptr_t)code & 15) || (c_cs & 15)) return -2;   // 16-byte block stores
    FwdArgs a{a1, a1_cs, wp, wp_cs, bias, b_cs, a2, a2_cs, code, c_cs, K, B};
    if (want_small(flags)) {
        dim3 grid(cdiv(K, 8) * 8 * B);
        if (KS == 5) LAUNCH_SMALL(conv2_fwd_f32_small_kernel<5>, grid, dim3(256), stream, a);
        else if (KS == 3) LAUNCH_SMALL(conv2_fwd_f32_small_kernel<3>, grid, dim3(256), stream, a);
        else return -3;
        P2PFL_RETURN_LAST_ERROR();
    }
    dim3 grid(cdiv(K, 8) * 8 * cdiv(B, 2));
    if (KS == 5) hipLaunchKernelGGL(conv2_fwd_f32_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL(conv2_fwd_f32_kernel<3>, grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

int conv2_dgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* wp, long wp_cs,
                    float* da1, long da1_cs, int K, int B, int KS, int flags, hipStream_t stream) {
    if (K <= 0 || B <= 0) return -1;
    if ((wp_cs & 3) || ((uintptr_t)wp & 15) || (da1_cs & 3) || ((uintptr_t)da1 & 15)) return -2;
    DgradArgs a{g, g_cs, code, c_cs, wp, wp_cs, da1, da1_cs, K, B};
    dim3 grid(cdiv(K, 8) * 8 * B);
    const bool small = want_small(flags);
    if (KS == 5 && small) LAUNCH_SMALL(conv2_dgrad_f32_small_kernel<5>, grid, dim3(256), stream, a);
    else if (KS == 5) hipLaunchKernelGGL(conv2_dgrad_f32_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 3 && small) LAUNCH_SMALL(conv2_dgrad_f32_small_kernel<3>, grid, dim3(256), stream, a);
    else if (KS == 3) hipLaunchKernelGGL(conv2_dgrad_f32_kernel<3>, grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

long conv2_wgrad_f32_scratch(int K, int B, int KS) {      // (enough for either variant)
    if (K <= 0 || B <= 0) return 0;
    int np, ns, ns2;
    wgrad_split_f32(K, B, KS, num_cus(), false, np, ns);
    wgrad_split_f32(K, B, KS, num_cus(), true, np, ns2);
    ns = ns > ns2 ? ns : ns2;
    return ns > 1 ? (long)K * ns * (KS * KS * COUT * CIN + COUT) : 0;
}

int conv2_wgrad_f32(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* a1, long a1_cs, float* dw,
                    long dw_cs, float* db, long db_cs, float* scratch, long scratch_floats, int K, int B, int KS,
                    int flags, const P2pflOpt* opt, hipStream_t stream) {
    if (K <= 0 || B <= 0) return -1;
    if (((uintptr_t)a1 & 15) || (a1_cs & 3)) return -2;
    const bool small = want_small(flags);
    const bool fold = opt != nullptr && opt->kind != 0;
    int np, ns;
    wgrad_split_f32(K, B, KS, num_cus(), small, np, ns);
    const long pw = (long)KS * KS * COUT * CIN + COUT;
    if (scratch == nullptr || scratch_floats < (long)K * ns * pw) ns = 1;
    if (fold && (scratch == nullptr || scratch_floats < (long)K * ns * pw)) return -5;
    WgradArgs a{g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, K, B, (ns > 1 || fold) ? scratch : nullptr, ns,
                make_opt_epi(opt)};
    if (KS == 5) launch_wgrad<5>(a, np, small, stream);
    else if (KS == 3) launch_wgrad<3>(a, np, small, stream);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}
