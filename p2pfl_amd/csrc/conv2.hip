// Stage-2 convolution (C_in = 32 -> C_out = 64, 5x5 or 3x3 "same", + ReLU + 2x2 max-pool) of the
// MNIST / FEMNIST CNNs for gfx950, forward and data gradient.
//
// Design: "resident image, shifted reads". A block keeps the (zero-padded) input image of its
// samples in LDS as channel-last bf16 -- [y][x][c], one pixel = one 64-B (fwd, 32 ch) or 128-B
// (dgrad, 64 ch) run -- so the MFMA A fragment of output pixel p for filter tap (ky,kx) and channel
// block c0 is ONE ds_read_b128 at  addr(p) + (ky*row + kx*pixel) + c0 : the im2col matrix is never
// formed, not even in LDS, and every tap costs a constant address offset. The K loop walks the taps;
// per tap the [C_out][C_in] weight slice (4 KB of a bf16, tap-major copy made once per step by
// p2pfl_conv2_prep) is double-buffered through LDS.
//
//   fwd  : M = pool-ordered pre-pool pixels m = (window*4 + q); the four rows of a pool window are
//          four consecutive accumulator registers of one lane -> bias + ReLU + max-pool + arg-max
//          code in registers. 2 samples / block, 13 row tiles over 4 waves, both 32-col halves per wave.
//   dgrad: image = pooled gradient routed to the arg-max pixel while staging (75 % zeros are never
//          read from HBM), M = natural pixels of one sample, N = C_in, taps mirrored.
//
// LDS layout of the 64-B-pixel images (forward, split data gradient): ds_read_b128 serves a wave in four 16-lane
// groups over the 16 slots (16 B) of the 256-B bank row. The row stride is 128 mod 256 bytes and the four 16-byte
// channel chunks of a pixel are ROTATED by a function of the pixel position, chosen per kernel so that the 16 lanes of
// every group hit 16 different slots for every tap (tools/probe/lds_conflict_sim.py: 4.0 cycles per read, was 6.6
// forward / 8.3 split data gradient with the padded strides; SQ_LDS_BANK_CONFLICT 29 % / 48 % of the LDS cycles):
//   forward : a row tile = the 7 pool windows of ONE window row, windows placed on the 8 four-lane slots so that each
//             lane group holds window columns 0-3 or 4-6 (distinct mod 4); chunk rotation = (X >> 1)
//   dgrad   : natural pixel order; chunk rotation = ((14 Y + X) >> 2)  (= (p + tap offset) >> 2 for output pixel p)
// Blocks of one client are mapped to one XCD so its weights stay in that XCD's L2.
//
// Reference layers: fedstellar/learning/pytorch/mnist/models/cnn.py:138-142,152-154 and
// fedstellar/learning/pytorch/femnist/models/cnn.py:137-139,147-148.
#include <cstdlib>
#include <type_traits>

#include "conv_f32.h"
#include "opt_epi.h"

namespace {

constexpr int P1 = 14, P2 = 7, CELLS1 = 196, CELLS2 = 49;
constexpr int CIN = 32, COUT = 64;

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int mfma_row_(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int KS>
struct Geo {
    static constexpr int PAD = KS / 2;
    static constexpr int IW = P1 + 2 * PAD;
    static constexpr int T = KS * KS;
    // forward / split-gradient image: 64 B / pixel, row stride = 128 (mod 256)
    static constexpr int XRS = IW * 64 + (((128 - IW * 64) % 256) + 256) % 256;
    static constexpr int XSS = IW * XRS;
    // gradient image: 128 B + 16 pad / pixel; row pad makes the 14-pixel row wrap look like +1 pixel
    static constexpr int GPS = 144;
    static constexpr int GRS = IW * GPS + 16 * ((((P1 - IW) * 9) % 16 + 16) % 16);
    static constexpr int GSS = IW * GRS;
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
// weight prep: W[n][c][t] fp32 -> Wf[t][n][c] (fwd B operand) and Wd[t'][c][n] (dgrad B operand,
// t' = mirrored tap) in bf16
// ---------------------------------------------------------------------------------------
template <int KS, int NS>
__global__ __launch_bounds__(256) void conv2_prep_kernel(const float* w, long w_cs, __bf16* wp, long wp_cs) {
    constexpr int T = KS * KS;
    constexpr int PLANE = 2 * T * COUT * CIN;     // one term: forward copy, then data-gradient copy
    const int c = blockIdx.y;
    const float* src = w + c * w_cs;
    __bf16* wf = wp + c * wp_cs;
    __bf16* wd = wf + T * COUT * CIN;
    // output-ordered: coalesced bf16 writes, strided (cache-resident) fp32 reads - the input-ordered form
    // scattered two 2-byte stores per element 4 KB apart (175 -> 101 us for 1000 clients 3x3, 64 -> 41 us for 100 clients 5x5)
    for (int o = blockIdx.x * 256 + threadIdx.x; o < COUT * CIN * T; o += gridDim.x * 256) {
        {   // wf[(t * COUT + n) * CIN + ci]
            const int ci = o % CIN, tn = o / CIN;
            const int n = tn % COUT, t = tn / COUT;
            float r = src[(n * CIN + ci) * T + t];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const __bf16 v = (__bf16)r;
                wf[s * PLANE + o] = v;
                r -= (float)v;
            }
        }
        {   // wd[((T - 1 - t) * CIN + ci) * COUT + n]
            const int n = o % COUT, tc = o / COUT;
            const int ci = tc % CIN, t = T - 1 - tc / CIN;
            float r = src[(n * CIN + ci) * T + t];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const __bf16 v = (__bf16)r;
                wd[s * PLANE + o] = v;
                r -= (float)v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
struct FwdArgs {
    const float* a1;   long a1_cs;     // [K][B][32][196] fp32
    const __bf16* wp;  long wp_cs;     // prepared weights
    const float* bias; long b_cs;      // [K][64]
    float* a2;         long a2_cs;     // [K][B][64][49]
    uint8_t* code;     long c_cs;
    int K, B;
    int ipb;                           // images per block of conv2_fwd_dense_kernel (0 = 1)
};

// NS = bf16 terms per operand (1: bf16 mode; 3: exact fp32 split, six MFMAs per fragment pair - PREC_F32_SPLIT).
// NS == 1: 2 samples / block, double-buffered weight slice. NS == 3: the three term images of ONE sample and a
// single-buffered slice (78 KB: two blocks per CU), 7 row tiles over 4 waves.
template <int KS, int NS>
__global__ __launch_bounds__(256, 2) void conv2_fwd_kernel(FwdArgs a) {
    using G = Geo<KS>;
    constexpr int S = NS == 1 ? 2 : 1;         // samples per block
    constexpr int NBUF = NS == 1 ? 2 : 1;
    constexpr int NTILE = S * P2;              // row tiles: one window row (7 windows on 8 slots) each
    constexpr int MT = (NTILE + 3) / 4;        // tiles per wave
    constexpr unsigned SLOT_CX = 0x43261540u;  // window column of slot j (nibble j); slot 7 repeats column 4 (unused)
    constexpr int IMG = S * G::XSS;            // bytes of one term's images
    constexpr int WSL = COUT * 80;             // bytes of one term's weight slice
    constexpr int PLANE = 2 * G::T * COUT * CIN;
    __shared__ __attribute__((aligned(16))) unsigned char ximg[NS * IMG];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[NBUF][NS * WSL];

    int c, sub;
    if (!xcd_map(cdiv(a.B, S), a.K, c, sub)) return;
    const int b0 = sub * S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- stage the padded images (fp32 NCHW -> bf16 terms, channel-last)
    const float* a1 = a.a1 + c * a.a1_cs;
    for (int task = tid; task < S * G::IW * G::IW * 4; task += 256) {
        const int g = task & 3, pp = task >> 2;
        const int s = pp / (G::IW * G::IW), rr = pp % (G::IW * G::IW);
        const int Y = rr / G::IW, X = rr % G::IW;
        const int y = Y - G::PAD, x = X - G::PAD;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (y >= 0 && y < P1 && x >= 0 && x < P1 && b0 + s < a.B) {
            const float* src = a1 + ((size_t)(b0 + s) * CIN + 8 * g) * CELLS1 + y * P1 + x;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[i * CELLS1];
        }
        bf16x8 tv[NS];
        split_bf16x8<NS>(v, tv);
#pragma unroll
        for (int e = 0; e < NS; ++e)
            *reinterpret_cast<bf16x8*>(ximg + e * IMG + s * G::XSS + Y * G::XRS + X * 64 + ((g + (X >> 1)) & 3) * 16) = tv[e];
    }
    const __bf16* wf = a.wp + c * a.wp_cs;
#pragma unroll
    for (int e = 0; e < NS; ++e) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wf + e * PLANE + tid * 8);
        *reinterpret_cast<bf16x8*>(wbuf[0] + e * WSL + (tid >> 2) * 80 + (tid & 3) * 16) = w0;
    }
    __syncthreads();

    // ---- per-lane image offsets of my row tiles (tiles past the end recompute window 0: an unguarded MFMA
    // loop - a per-tile guard compiles to a branch plus accumulator copies around every MFMA)
    const int ntile = (NTILE - wave + 3) / 4;          // tiles wave, wave+4, ...
    const int xl = 2 * (int)((SLOT_CX >> (4 * (r >> 2))) & 7u) + (r & 1);     // my pixel column, the same in every tile
    int abase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int tl = wave + 4 * i;
        if (tl >= NTILE) tl = 0;
        const int y = 2 * (tl % P2) + ((r >> 1) & 1);
        abase[i] = (tl / P2) * G::XSS + y * G::XRS + xl * 64;
    }
    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int ky = 0, kx = 0;
#pragma unroll 1
    for (int t = 0; t < G::T; ++t) {
        bf16x8 wnext[NS];
        if (t + 1 < G::T) {
#pragma unroll
            for (int e = 0; e < NS; ++e)
                wnext[e] = *reinterpret_cast<const bf16x8*>(wf + e * PLANE + (t + 1) * COUT * CIN + tid * 8);
        }
        const unsigned char* wb = wbuf[NBUF == 2 ? (t & 1) : 0] + r * 80 + h * 16;
        const int rot = h + ((xl + kx) >> 1);            // chunk rotation of the pixel this lane reads for this tap
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int toff = ky * G::XRS + kx * 64 + ((rot + 2 * ks) & 3) * 16;
            bf16x8 bf[NS][2];
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                bf[e][0] = *reinterpret_cast<const bf16x8*>(wb + e * WSL + ks * 32);
                bf[e][1] = *reinterpret_cast<const bf16x8*>(wb + e * WSL + 32 * 80 + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                bf16x8 af[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    af[e] = *reinterpret_cast<const bf16x8*>(ximg + e * IMG + abase[i] + toff);
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bf[eb][0], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bf[eb][1], acc[i][1], 0, 0, 0);
                    }
            }
        }
        if (NBUF == 1) __syncthreads();                  // every wave finished reading the single slice
        if (t + 1 < G::T) {
#pragma unroll
            for (int e = 0; e < NS; ++e)
                *reinterpret_cast<bf16x8*>(wbuf[NBUF == 2 ? ((t + 1) & 1) : 0] + e * WSL + (tid >> 2) * 80 +
                                           (tid & 3) * 16) = wnext[e];
        }
        __syncthreads();
        if (++kx == KS) { kx = 0; ++ky; }
    }

    // ---- epilogue: bias + ReLU + 2x2 max-pool + arg-max code from the accumulators. A lane owns an output
    // channel, whose pooled cells are 196 B apart in memory: direct stores would be 208 fully scattered
    // 4-byte / 1-byte store instructions per wave. The block's outputs (S samples x [64][49], one contiguous
    // 25 KB piece of a2 and 6 KB of codes) are assembled in the now idle image buffer and written as 16-byte
    // pieces per lane instead.
    static_assert(NS * IMG >= S * COUT * CELLS2 * 5, "output staging must fit into the image buffer");
    float* out_s = reinterpret_cast<float*>(ximg);
    uint8_t* code_s = ximg + S * COUT * CELLS2 * 4;
    const float* bias = a.bias + c * a.b_cs;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i >= ntile) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = j * 32 + r;
            const float bv = bias[n];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int slot = 2 * g + h, tl = wave + 4 * i;
                if (slot == 7) continue;                   // (the eighth slot of a 7-window row)
                const int W = (tl / P2) * CELLS2 + (tl % P2) * P2 + (int)((SLOT_CX >> (4 * slot)) & 7u);
                float best = acc[i][j][4 * g]; int q = 0;
                if (acc[i][j][4 * g + 1] > best) { best = acc[i][j][4 * g + 1]; q = 1; }
                if (acc[i][j][4 * g + 2] > best) { best = acc[i][j][4 * g + 2]; q = 2; }
                if (acc[i][j][4 * g + 3] > best) { best = acc[i][j][4 * g + 3]; q = 3; }
                best += bv;
                if (!(best > 0.f)) { best = 0.f; q = 4; }
                const int o = ((W / CELLS2) * COUT + n) * CELLS2 + W % CELLS2;     // == the global layout of the block
                out_s[o] = best;
                code_s[o] = (uint8_t)q;
            }
        }
    }
    __syncthreads();
    const int nvalid = min(S, a.B - b0) * COUT * CELLS2;                          // floats (= code bytes) to write
    float* a2 = a.a2 + c * a.a2_cs + (size_t)b0 * COUT * CELLS2;
    uint8_t* code = a.code + c * a.c_cs + (size_t)b0 * COUT * CELLS2;
    for (int i = tid; i < nvalid / 4; i += 256)
        reinterpret_cast<f32x4*>(a2)[i] = reinterpret_cast<const f32x4*>(out_s)[i];
    for (int i = tid; i < nvalid / 16; i += 256)
        reinterpret_cast<f32x4*>(code)[i] = reinterpret_cast<const f32x4*>(code_s)[i];
}

// ---------------------------------------------------------------------------------------
// forward, exact-split policy, DENSE window order.
//
// The kernel above gives a row tile to one window row: 7 windows on 8 slots, 7 tiles issued as 8 over the four waves -
// 196 of 256 issued rows are real (76.6 %), and its tap loop is ~80 % matrix-pipe-bound. Here the 49 pool windows of the
// sample are numbered densely (W = 7 wy + wx, rows m = 4 W + q): six full 32-row tiles (windows 0..47) and ONE window
// left over. The twelve (tile, 32-column half) units are dealt three to a wave, so a wave reads the weight fragments of
// one column half only, and the left-over window runs on v_mfma_f32_16x16x32_bf16 (rows 0..3 real, one 16-column tile
// per wave, all 32 input channels in one instruction). Matrix-pipe time per tap and wave: 36 x 32 + 6 x 16 = 1248
// cycles against 48 x 32 = 1536.
// LDS: the chunk rotation of a pixel is ((X >> 1) - (Y >> 1)) & 3 = (W + const) & 3 for the pixels that the four windows
// of a 16-lane group read for one tap - four consecutive windows, four different rotations, also across a window-row
// wrap. Not conflict-free like the window-row layout: for taps with odd ky the two rows of a window fall into different
// rotation classes (model: 1.33 - 1.4 LDS cycles per fragment read, no rotation that is linear in the pixel position does
// better; SQ_LDS_BANK_CONFLICT 30 % of the LDS cycles against 16 %, profiles/r3_final_pmc_summary.csv). The kernel trades
// those cycles for a fifth of the matrix-pipe time and comes out 10 % ahead.
// ---------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, 2) void conv2_fwd_dense_kernel(FwdArgs a) {
    using G = Geo<KS>;
    constexpr int NS = 3;
    constexpr int IMG = G::XSS;                // bytes of one term's image
    constexpr int WSL = COUT * 80;             // bytes of one term's weight slice
    constexpr int PLANE = 2 * G::T * COUT * CIN;
    __shared__ __attribute__((aligned(16))) unsigned char ximg[NS * IMG];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[NS * WSL];

    const int ipb = a.ipb > 0 ? a.ipb : 1;
    int c, sub;
    if (!xcd_map(cdiv(a.B, ipb), a.K, c, sub)) return;
    const int b_begin = sub * ipb, b_end = min(a.B, b_begin + ipb);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- staging of one sample in two halves: global loads into registers (issued BEFORE the previous sample's tap loop,
    // so they fly under its MFMAs), then split + LDS stores (fp32 NCHW -> bf16 terms, channel-last, rotated chunks)
    const float* a1 = a.a1 + c * a.a1_cs;
    const __bf16* wf = a.wp + c * a.wp_cs;
    constexpr int NTASK = G::IW * G::IW * 4, NIT = (NTASK + 255) / 256;
    float sv[NIT][8];
    bf16x8 w0[NS];
    auto prefetch = [&](int b) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int task = tid + 256 * u;
            const int g = task & 3, rr = task >> 2;
            const int y = rr / G::IW - G::PAD, x = rr % G::IW - G::PAD;
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[u][i] = 0.f;
            if (task < NTASK && y >= 0 && y < P1 && x >= 0 && x < P1) {
                const float* src = a1 + ((size_t)b * CIN + 8 * g) * CELLS1 + y * P1 + x;
#pragma unroll
                for (int i = 0; i < 8; ++i) sv[u][i] = src[i * CELLS1];
            }
        }
#pragma unroll
        for (int e = 0; e < NS; ++e) w0[e] = *reinterpret_cast<const bf16x8*>(wf + e * PLANE + tid * 8);
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int task = tid + 256 * u;
            if (task < NTASK) {
                const int g = task & 3, rr = task >> 2;
                const int Y = rr / G::IW, X = rr % G::IW;
                bf16x8 tv[NS];
                split_bf16x8<NS>(sv[u], tv);
                const int pos = (g + (X >> 1) + 3 * (Y >> 1)) & 3;
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    *reinterpret_cast<bf16x8*>(ximg + e * IMG + Y * G::XRS + X * 64 + pos * 16) = tv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < NS; ++e)
            *reinterpret_cast<bf16x8*>(wbuf + e * WSL + (tid >> 2) * 80 + (tid & 3) * 16) = w0[e];
    };

    // ---- my units: column half jc, row tiles 3 (wave >> 1) + {0, 1, 2}; my pixel of each tile (padded coordinates of
    // tap (0, 0)): window 8 tile + (r >> 2), corner q = r & 3
    const int jc = wave & 1, t0 = 3 * (wave >> 1);
    int px[3], py[3], abase[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int Wd = 8 * (t0 + i) + (r >> 2);
        py[i] = 2 * (Wd / P2) + ((r >> 1) & 1);
        px[i] = 2 * (Wd % P2) + (r & 1);
        abase[i] = py[i] * G::XRS + px[i] * 64;
    }
    // the left-over window (48 = (6, 6)) on the 16-row instruction: lane -> row = lane & 15 (corner (lane & 3), rows >= 4
    // repeat it), channel chunk lane >> 4; B: column 16 wave + (lane & 15)
    const int lq = lane & 3, lg = lane >> 4;
    const int lpy = 12 + (lq >> 1), lpx = 12 + (lq & 1);
    const int lbase = lpy * G::XRS + lpx * 64;
    prefetch(b_begin);
#pragma unroll 1
    for (int b = b_begin; b < b_end; ++b) {
        commit();
        __syncthreads();
        if (b + 1 < b_end) prefetch(b + 1);
        f32x16 acc[3];
        f32x4 accl = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
        for (int i = 0; i < 3; ++i)
    #pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        int ky = 0, kx = 0;
    #pragma unroll 1
        for (int t = 0; t < G::T; ++t) {
            bf16x8 wnext[NS];
            if (t + 1 < G::T) {
    #pragma unroll
                for (int e = 0; e < NS; ++e)
                    wnext[e] = *reinterpret_cast<const bf16x8*>(wf + e * PLANE + (t + 1) * COUT * CIN + tid * 8);
            }
            const unsigned char* wb = wbuf + (32 * jc + r) * 80 + h * 16;
            const int tap_off = ky * G::XRS + kx * 64;
            int rot[3];
    #pragma unroll
            for (int i = 0; i < 3; ++i) rot[i] = ((px[i] + kx) >> 1) + 3 * ((py[i] + ky) >> 1) + h;
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bf[NS];
    #pragma unroll
                for (int e = 0; e < NS; ++e) bf[e] = *reinterpret_cast<const bf16x8*>(wb + e * WSL + ks * 32);
    #pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int off = abase[i] + tap_off + ((rot[i] + 2 * ks) & 3) * 16;
                    bf16x8 af[NS];
    #pragma unroll
                    for (int e = 0; e < NS; ++e) af[e] = *reinterpret_cast<const bf16x8*>(ximg + e * IMG + off);
    #pragma unroll
                    for (int ea = 0; ea < NS; ++ea)
    #pragma unroll
                        for (int eb = 0; eb < NS - ea; ++eb)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bf[eb], acc[i], 0, 0, 0);
                }
            }
            {   // left-over window: all 32 channels in one instruction, 16 columns per wave
                const int lrot = ((lpx + kx) >> 1) + 3 * ((lpy + ky) >> 1) + lg;
                const int off = lbase + tap_off + (lrot & 3) * 16;
                const unsigned char* wl = wbuf + (16 * wave + (lane & 15)) * 80 + lg * 16;
                bf16x8 af[NS], bl[NS];
    #pragma unroll
                for (int e = 0; e < NS; ++e) {
                    af[e] = *reinterpret_cast<const bf16x8*>(ximg + e * IMG + off);
                    bl[e] = *reinterpret_cast<const bf16x8*>(wl + e * WSL);
                }
    #pragma unroll
                for (int ea = 0; ea < NS; ++ea)
    #pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
                        accl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ea], bl[eb], accl, 0, 0, 0);
            }
            __syncthreads();                                 // every wave finished reading the single slice
            if (t + 1 < G::T) {
    #pragma unroll
                for (int e = 0; e < NS; ++e)
                    *reinterpret_cast<bf16x8*>(wbuf + e * WSL + (tid >> 2) * 80 + (tid & 3) * 16) = wnext[e];
            }
            __syncthreads();
            if (++kx == KS) { kx = 0; ++ky; }
        }

        // ---- epilogue: bias + ReLU + 2x2 max-pool + arg-max code from the accumulators, assembled in the idle image buffer
        // ([64][49] floats + codes = the sample's contiguous piece of a2 / code) and written as 16-byte pieces
        static_assert(NS * IMG >= COUT * CELLS2 * 5, "output staging must fit into the image buffer");
        float* out_s = reinterpret_cast<float*>(ximg);
        uint8_t* code_s = ximg + COUT * CELLS2 * 4;
        const float* bias = a.bias + c * a.b_cs;
        auto pool = [&](float v0, float v1, float v2, float v3, float bv, int n, int W) {
            float best = v0; int q = 0;
            if (v1 > best) { best = v1; q = 1; }
            if (v2 > best) { best = v2; q = 2; }
            if (v3 > best) { best = v3; q = 3; }
            best += bv;
            if (!(best > 0.f)) { best = 0.f; q = 4; }
            out_s[n * CELLS2 + W] = best;
            code_s[n * CELLS2 + W] = (uint8_t)q;
        };
        {
            const int n = 32 * jc + r;
            const float bv = bias[n];
    #pragma unroll
            for (int i = 0; i < 3; ++i)
    #pragma unroll
                for (int g = 0; g < 4; ++g)
                    pool(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3], bv, n, 8 * (t0 + i) + 2 * g + h);
        }
        if (lane < 16) {     // 16 x 16 accumulator: lane = column, rows 4 (lane >> 4) + reg -> lanes 0..15 hold window 48
            const int n = 16 * wave + lane;
            pool(accl[0], accl[1], accl[2], accl[3], bias[n], n, CELLS2 - 1);
        }
        __syncthreads();
        float* a2 = a.a2 + c * a.a2_cs + (size_t)b * COUT * CELLS2;
        uint8_t* code = a.code + c * a.c_cs + (size_t)b * COUT * CELLS2;
        for (int i = tid; i < COUT * CELLS2 / 4; i += 256)
            reinterpret_cast<f32x4*>(a2)[i] = reinterpret_cast<const f32x4*>(out_s)[i];
        for (int i = tid; i < COUT * CELLS2 / 16; i += 256)
            reinterpret_cast<f32x4*>(code)[i] = reinterpret_cast<const f32x4*>(code_s)[i];
        __syncthreads();               // the output staging is read: the image buffer may take the next sample
    }
}

// ---------------------------------------------------------------------------------------
// data gradient
// ---------------------------------------------------------------------------------------
struct DgradArgs {
    const float* g;    long g_cs;      // da2 [K][B][64][49]
    const uint8_t* code; long c_cs;
    const __bf16* wp;  long wp_cs;
    float* da1;        long da1_cs;    // [K][B][32][196]
    int K, B;
};

template <int KS>
__global__ __launch_bounds__(256, 2) void conv2_dgrad_kernel(DgradArgs a) {
    using G = Geo<KS>;
    constexpr int NTILE = (CELLS1 + 31) / 32;   // 7 row tiles of 32 pixels
    constexpr int MT = (NTILE + 3) / 4;         // (2,2,2,1)
    constexpr int WRS = 144;                    // LDS row of the weight slice: 64 n bf16 + 16 B pad
    __shared__ __attribute__((aligned(16))) unsigned char gimg[G::GSS];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[2][CIN * WRS];

    int c, b;
    if (!xcd_map(a.B, a.K, c, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // ---- zero the image (halo + pixels of dead windows), then route the pooled gradient
    for (int i = tid; i < G::GSS / 16; i += 256)
        *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    const __bf16* wd = a.wp + c * a.wp_cs + G::T * COUT * CIN;
    {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wd + tid * 8);
        *reinterpret_cast<bf16x8*>(wbuf[0] + (tid >> 3) * WRS + (tid & 7) * 16) = w0;
    }
    __syncthreads();
    const float* gsrc = a.g + c * a.g_cs + (size_t)b * COUT * CELLS2;
    const uint8_t* csrc = a.code + c * a.c_cs + (size_t)b * COUT * CELLS2;
    for (int task = tid; task < CELLS2 * 8; task += 256) {
        const int cg = task / CELLS2, cell = task % CELLS2;
        float gv[8]; int qc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gv[i] = gsrc[(cg * 8 + i) * CELLS2 + cell];
            qc[i] = csrc[(cg * 8 + i) * CELLS2 + cell];
        }
        const int Y = 2 * (cell / P2) + G::PAD, X = 2 * (cell % P2) + G::PAD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (qc[i] == q) ? gv[i] : 0.f;
            *reinterpret_cast<bf16x8*>(gimg + (Y + (q >> 1)) * G::GRS + (X + (q & 1)) * G::GPS + cg * 16) =
                pack_bf16x8(v);
        }
    }
    __syncthreads();

    const int ntile = (NTILE - wave + 3) / 4;
    int abase[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int p = (wave + 4 * i) * 32 + r;
        if (p >= CELLS1) p = 0;
        abase[i] = (p / P1) * G::GRS + (p % P1) * G::GPS + h * 16;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    int ky = 0, kx = 0;   // mirrored tap index: image offset (ky,kx), weights of tap (KS-1-ky, KS-1-kx)
#pragma unroll 1
    for (int t = 0; t < G::T; ++t) {
        bf16x8 wnext;
        if (t + 1 < G::T) wnext = *reinterpret_cast<const bf16x8*>(wd + (t + 1) * COUT * CIN + tid * 8);
        const unsigned char* wb = wbuf[t & 1] + r * WRS + h * 16;
        const int toff = ky * G::GRS + kx * G::GPS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wb + ks * 32);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < ntile) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(gimg + abase[i] + toff + ks * 32);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
                }
            }
        }
        if (t + 1 < G::T)
            *reinterpret_cast<bf16x8*>(wbuf[(t + 1) & 1] + (tid >> 3) * WRS + (tid & 7) * 16) = wnext;
        __syncthreads();
        if (++kx == KS) { kx = 0; ++ky; }
    }

    // ---- epilogue: lane = input channel, registers 4g..4g+3 = four consecutive pixels -> float4
    // (assembling the sample in LDS for address-ordered stores was measured: no gain over these 16-byte pieces)
    float* dst = a.da1 + c * a.da1_cs + ((size_t)b * CIN + r) * CELLS1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i >= ntile) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int p = (wave + 4 * i) * 32 + 8 * g + 4 * h;
            if (p < CELLS1)
                *reinterpret_cast<float4*>(dst + p) =
                    make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
    }
}

// Data gradient with NS-term split operands (PREC_F32_SPLIT). The 64 gradient channels go through the matrix cores
// in two passes of 32, so the three term images of a pass have the forward kernel's geometry (64-B pixels, 63 KB)
// and two blocks share a CU; the accumulators carry over the passes.
template <int KS, int NS>
__global__ __launch_bounds__(256, 2) void conv2_dgrad_split_kernel(DgradArgs a) {
    using G = Geo<KS>;
    constexpr int NTILE = (CELLS1 + 31) / 32;   // 7 row tiles of 32 pixels
    constexpr int MT = (NTILE + 3) / 4;         // (2,2,2,1)
    constexpr int WRS = 80;                     // LDS row of the weight slice: 32 n bf16 + 16 B pad
    constexpr int IMG = G::XSS, WSL = CIN * WRS;
    constexpr int PLANE = 2 * G::T * COUT * CIN;
    __shared__ __attribute__((aligned(16))) unsigned char gimg[NS * IMG];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[2][NS * WSL];

    int c, b;
    if (!xcd_map(a.B, a.K, c, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const __bf16* wd = a.wp + c * a.wp_cs + G::T * COUT * CIN;
    const float* gsrc = a.g + c * a.g_cs + (size_t)b * COUT * CELLS2;
    const uint8_t* csrc = a.code + c * a.c_cs + (size_t)b * COUT * CELLS2;

    // weight slice of (tap, pass): per term 32 rows (ci) x 32 n (co of this pass) = 128 16-byte pieces
    constexpr int WPIECES = NS * 128, WROUNDS = (WPIECES + 255) / 256;
    auto wload = [&](int t, int pass, bf16x8 (&w)[WROUNDS]) {
#pragma unroll
        for (int u = 0; u < WROUNDS; ++u) {
            const int q = tid + 256 * u;
            if (q < WPIECES) {
                const int e = q >> 7, rem = q & 127;
                w[u] = *reinterpret_cast<const bf16x8*>(wd + e * PLANE + t * COUT * CIN + (rem >> 2) * COUT + pass * 32 +
                                                        (rem & 3) * 8);
            }
        }
    };
    auto wstore = [&](int buf, const bf16x8 (&w)[WROUNDS]) {
#pragma unroll
        for (int u = 0; u < WROUNDS; ++u) {
            const int q = tid + 256 * u;
            if (q < WPIECES) {
                const int e = q >> 7, rem = q & 127;
                *reinterpret_cast<bf16x8*>(wbuf[buf] + e * WSL + (rem >> 2) * WRS + (rem & 3) * 16) = w[u];
            }
        }
    };

    const int ntile = (NTILE - wave + 3) / 4;
    int abase[MT], pl[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int p = (wave + 4 * i) * 32 + r;
        if (p >= CELLS1) p = 0;
        pl[i] = p + 4 * h;                       // (rotation of the pixel read for a tap: ((p + tap) >> 2) + h + 2 ks)
        abase[i] = (p / P1) * G::XRS + (p % P1) * 64;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();                 // all reads of the previous pass's images and slices are done
        // ---- zero the term images (halo + pixels of dead windows), then route this pass's pooled gradient
        for (int i = tid; i < NS * IMG / 16; i += 256)
            *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            bf16x8 w0[WROUNDS];
            wload(0, pass, w0);
            wstore(0, w0);
        }
        __syncthreads();
        for (int task = tid; task < CELLS2 * 4; task += 256) {
            const int cg = task / CELLS2, cell = task % CELLS2;
            float gv[8]; int qc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                gv[i] = gsrc[(pass * 32 + cg * 8 + i) * CELLS2 + cell];
                qc[i] = csrc[(pass * 32 + cg * 8 + i) * CELLS2 + cell];
            }
            bf16x8 tv[NS];
            split_bf16x8<NS>(gv, tv);
            const int Y = 2 * (cell / P2) + G::PAD, X = 2 * (cell % P2) + G::PAD;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    bf16x8 v;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (qc[i] == q) ? tv[e][i] : (__bf16)0.f;
                    const int Yq = Y + (q >> 1), Xq = X + (q & 1);
                    *reinterpret_cast<bf16x8*>(gimg + e * IMG + Yq * G::XRS + Xq * 64 +
                                               ((cg + ((P1 * Yq + Xq) >> 2)) & 3) * 16) = v;
                }
            }
        }
        __syncthreads();

        int ky = 0, kx = 0;   // mirrored tap index: image offset (ky,kx), weights of tap (KS-1-ky, KS-1-kx)
#pragma unroll 1
        for (int t = 0; t < G::T; ++t) {
            bf16x8 wnext[WROUNDS];
            if (t + 1 < G::T) wload(t + 1, pass, wnext);
            const unsigned char* wb = wbuf[t & 1] + r * WRS + h * 16;
            const int toff = ky * G::XRS + kx * 64, tapl = P1 * ky + kx;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bfr[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) bfr[e] = *reinterpret_cast<const bf16x8*>(wb + e * WSL + ks * 32);
                bf16x8 af[MT][NS];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int off = abase[i] + toff + ((((pl[i] + tapl) >> 2) + 2 * ks) & 3) * 16;
#pragma unroll
                    for (int e = 0; e < NS; ++e)
                        af[i][e] = *reinterpret_cast<const bf16x8*>(gimg + e * IMG + off);
                }
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ea], bfr[eb], acc[i], 0, 0, 0);
            }
            if (t + 1 < G::T) wstore((t + 1) & 1, wnext);
            __syncthreads();
            if (++kx == KS) { kx = 0; ++ky; }
        }
    }

    float* dst = a.da1 + c * a.da1_cs + ((size_t)b * CIN + r) * CELLS1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i >= ntile) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int p = (wave + 4 * i) * 32 + 8 * g + 4 * h;
            if (p < CELLS1)
                *reinterpret_cast<float4*>(dst + p) =
                    make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
        }
    }
}

// Data gradient, exact-split policy, without the phantom row tile: 196 pixels are six full 32-row tiles and four pixels.
// The kernel above issues 8 tiles (two per wave) for them. Here the twelve (tile, channel half of the pass) units are
// dealt three to a wave - wave w reduces channels 16 (w & 1) .. + 15 of the pass over tiles 3 (w >> 1) + {0, 1, 2}, the
// two waves of a tile triple add their accumulators through LDS at the end (fixed order) - and the four left-over pixels
// run on v_mfma_f32_16x16x32_bf16 (all 32 channels of the pass per instruction, one 16-column tile per wave: waves 0 / 1
// in the first pass, 2 / 3 in the second, summed at the end). Matrix-pipe time per tap, pass and wave: 18 x 32 (+ 6 x 16)
// cycles against 24 x 32.
template <int KS>
__global__ __launch_bounds__(256, 2) void conv2_dgrad_dense_kernel(DgradArgs a) {
    using G = Geo<KS>;
    constexpr int NS = 3;
    constexpr int WRS = 80;                     // LDS row of the weight slice: 32 n bf16 + 16 B pad
    constexpr int IMG = G::XSS, WSL = CIN * WRS;
    constexpr int PLANE = 2 * G::T * COUT * CIN;
    __shared__ __attribute__((aligned(16))) unsigned char gimg[NS * IMG];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf[2][NS * WSL];

    int c, b;
    if (!xcd_map(a.B, a.K, c, b)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const __bf16* wd = a.wp + c * a.wp_cs + G::T * COUT * CIN;
    const float* gsrc = a.g + c * a.g_cs + (size_t)b * COUT * CELLS2;
    const uint8_t* csrc = a.code + c * a.c_cs + (size_t)b * COUT * CELLS2;

    constexpr int WPIECES = NS * 128, WROUNDS = (WPIECES + 255) / 256;
    auto wload = [&](int t, int pass, bf16x8 (&w)[WROUNDS]) {
#pragma unroll
        for (int u = 0; u < WROUNDS; ++u) {
            const int q = tid + 256 * u;
            if (q < WPIECES) {
                const int e = q >> 7, rem = q & 127;
                w[u] = *reinterpret_cast<const bf16x8*>(wd + e * PLANE + t * COUT * CIN + (rem >> 2) * COUT + pass * 32 +
                                                        (rem & 3) * 8);
            }
        }
    };
    auto wstore = [&](int buf, const bf16x8 (&w)[WROUNDS]) {
#pragma unroll
        for (int u = 0; u < WROUNDS; ++u) {
            const int q = tid + 256 * u;
            if (q < WPIECES) {
                const int e = q >> 7, rem = q & 127;
                *reinterpret_cast<bf16x8*>(wbuf[buf] + e * WSL + (rem >> 2) * WRS + (rem & 3) * 16) = w[u];
            }
        }
    };

    const int ksw = wave & 1, t0 = 3 * (wave >> 1);
    int abase[3], pl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = (t0 + i) * 32 + r;
        pl[i] = p + 4 * h + 8 * ksw;             // (rotation of the pixel read for a tap: ((p + tap) >> 2) + h + 2 ks)
        abase[i] = (p / P1) * G::XRS + (p % P1) * 64;
    }
    // left-over pixels 192..195 (row 13, x = 10..13) on the 16-row instruction: lane -> row = lane & 15 (pixel 192 +
    // (lane & 3), rows >= 4 repeat it), channel chunk lane >> 4; B: input channel 16 (wave & 1) + (lane & 15)
    const int lp = 192 + (lane & 3), lg = lane >> 4;
    const int lbase = (lp / P1) * G::XRS + (lp % P1) * 64;
    f32x16 acc[3];
    f32x4 accl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();                 // all reads of the previous pass's images and slices are done
        // the routing below writes all four pixels of every window (zeros where the window's arg-max is elsewhere), i.e. the
        // whole interior: only the halo needs the zero fill, once
        if (pass == 0)
            for (int i = tid; i < NS * IMG / 16; i += 256)
                *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            bf16x8 w0[WROUNDS];
            wload(0, pass, w0);
            wstore(0, w0);
        }
        __syncthreads();
        for (int task = tid; task < CELLS2 * 4; task += 256) {
            const int cg = task / CELLS2, cell = task % CELLS2;
            float gv[8]; int qc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                gv[i] = gsrc[(pass * 32 + cg * 8 + i) * CELLS2 + cell];
                qc[i] = csrc[(pass * 32 + cg * 8 + i) * CELLS2 + cell];
            }
            bf16x8 tv[NS];
            split_bf16x8<NS>(gv, tv);
            const int Y = 2 * (cell / P2) + G::PAD, X = 2 * (cell % P2) + G::PAD;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    bf16x8 v;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (qc[i] == q) ? tv[e][i] : (__bf16)0.f;
                    const int Yq = Y + (q >> 1), Xq = X + (q & 1);
                    *reinterpret_cast<bf16x8*>(gimg + e * IMG + Yq * G::XRS + Xq * 64 +
                                               ((cg + ((P1 * Yq + Xq) >> 2)) & 3) * 16) = v;
                }
            }
        }
        // (the second pass's gradient loaded into registers before the first pass's tap loop: slower, 203.8 us against 194.4
        //  for 50 clients 5x5, 1357 against 1342 for 1000 clients 3x3)
        __syncthreads();
        const bool left = (wave >> 1) == pass;     // (wave-uniform) this wave owns the left-over pixels in this pass

        int ky = 0, kx = 0;   // mirrored tap index: image offset (ky,kx), weights of tap (KS-1-ky, KS-1-kx)
#pragma unroll 1
        for (int t = 0; t < G::T; ++t) {
            bf16x8 wnext[WROUNDS];
            if (t + 1 < G::T) wload(t + 1, pass, wnext);
            const unsigned char* wb = wbuf[t & 1] + r * WRS + h * 16 + ksw * 32;
            const int toff = ky * G::XRS + kx * 64, tapl = P1 * ky + kx;
            bf16x8 bfr[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) bfr[e] = *reinterpret_cast<const bf16x8*>(wb + e * WSL);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int off = abase[i] + toff + (((pl[i] + tapl) >> 2) & 3) * 16;
                bf16x8 af[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) af[e] = *reinterpret_cast<const bf16x8*>(gimg + e * IMG + off);
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bfr[eb], acc[i], 0, 0, 0);
            }
            if (left) {
                const int off = lbase + toff + ((((lp + tapl) >> 2) + lg) & 3) * 16;
                const unsigned char* wl = wbuf[t & 1] + (16 * ksw + (lane & 15)) * WRS + lg * 16;
                bf16x8 af[NS], bl[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    af[e] = *reinterpret_cast<const bf16x8*>(gimg + e * IMG + off);
                    bl[e] = *reinterpret_cast<const bf16x8*>(wl + e * WSL);
                }
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
                        accl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ea], bl[eb], accl, 0, 0, 0);
            }
            if (t + 1 < G::T) wstore((t + 1) & 1, wnext);
            __syncthreads();
            if (++kx == KS) { kx = 0; ++ky; }
        }
    }

    // ---- the odd waves hand their halves over (the loop's last barrier freed the images)
    float* red = reinterpret_cast<float*>(gimg);          // [2 pairs][3 tiles][16][64] + left-over [4 waves][4][64]
    float* redl = red + 2 * 3 * 16 * 64;
    static_assert(NS * IMG >= (2 * 3 * 16 * 64 + 4 * 4 * 64) * 4, "the exchange must fit into the image buffer");
    if (ksw) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(((wave >> 1) * 3 + i) * 16 + e) * 64 + lane] = acc[i][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) redl[(wave * 4 + e) * 64 + lane] = accl[e];
    __syncthreads();
    if (!ksw) {     // lane = input channel, registers 4g..4g+3 = four consecutive pixels -> float4
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
    } else if (wave == 1 || wave == 3) {
        // left-over pixels: 16 x 16 accumulators (lane & 15 = input channel of the wave's column tile, lanes 0..15 hold rows
        // 0..3 = pixels 192..195): wave 1 finishes column tile 0 (waves 0 + 2), wave 3 column tile 1 (waves 1 + 3)
        if (lane < 16) {
            const int j = wave >> 1;
            const int ci = 16 * j + lane;
            const float* p0 = redl + (j * 4) * 64 + lane;              // wave j     (first pass)
            const float* p1 = redl + ((2 + j) * 4) * 64 + lane;        // wave 2 + j (second pass)
            float* dst = a.da1 + c * a.da1_cs + ((size_t)b * CIN + ci) * CELLS1 + 192;
            *reinterpret_cast<float4*>(dst) = make_float4(p0[0] + p1[0], p0[64] + p1[64], p0[128] + p1[128], p0[192] + p1[192]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// weight gradient:  dW[n][c][t] = sum_{b,p} G[b][p][n] * X[b][p + t][c],   db[n] = sum G[b][p][n]
//
// GEMM per tap: M = C_out (2 row tiles), N = C_in (1 tile), K = pixels. Both operands need the pixel
// index along K while the LDS images are channel-last, so fragments are fetched with gfx950's
// transposing read ds_read_b64_tr_b16: one read hands every lane 4 consecutive pixels of its own
// channel, and a tap is again just a constant address offset on the X image.
// A block owns (client, 1/NP of the taps) and walks all samples of that client, so its accumulators
// are the final gradient: plain stores, no atomics (device-scope fp32 atomics across XCDs measured
// 20x slower than the whole GEMM), bitwise reproducible. 8 waves = 4 tap groups x 2 row tiles; a wave
// shares its G fragment across its taps; the bias gradient rides along as one more MFMA against a
// ones fragment. NP is picked by the launcher so that K * NP blocks fill the CUs.
// Image strides: 4 consecutive pixels fall into the four 64-B quarters of the 256-B bank row
// (X: 64-B pixels, row stride = 128 mod 256 so the 14-pixel row wrap keeps the pattern; G: 192-B
// pixels), i.e. the transposed reads are bank-conflict-free.
// ---------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lds0, const unsigned char* lds1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lds0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lds1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct WgradArgs {
    const float* g;    long g_cs;      // da2 [K][B][64][49]
    const uint8_t* code; long c_cs;
    const float* a1;   long a1_cs;     // [K][B][32][196]
    float* dw;         long dw_cs;     // [K][64][32*T]
    float* db;         long db_cs;     // [K][64]
    int K, B;
    float* part;       int NS;         // part != null: batch split, partials [K][NS][T*64*32 + 64] (see reduce kernel)
    OptEpi opt;                        // optimizer applied by the reduce kernel (kind 0: it writes dw / db)
};

template <int KS, int NP>
__global__ __launch_bounds__(512, 2) void conv2_wgrad_kernel(WgradArgs a) {
    using G = Geo<KS>;
    constexpr int T = G::T;
    constexpr int NTAP = ((T + NP - 1) / NP + 3) / 4;    // most taps per wave
    constexpr int XRS = G::IW * 64 + ((((P1 - G::IW) * 64) % 256) + 256) % 256;
    constexpr int XROWS = G::IW + 1;                     // +1 zero row: K padding (pixels 196..207) reads it
    constexpr int GPS = 192;
    constexpr int KSTEPS = (CELLS1 + 15) / 16;           // 13
    constexpr int NPIX = KSTEPS * 16;                    // 208 (last 12 stay zero)
    __shared__ __attribute__((aligned(16))) unsigned char ximg[XROWS * XRS];
    __shared__ __attribute__((aligned(16))) unsigned char gimg[NPIX * GPS];

    int c, sub;
    if (!xcd_map(NP * a.NS, a.K, c, sub)) return;
    const int tp = sub % NP, split = sub / NP;
    const int b_begin = split * a.B / a.NS, b_end = (split + 1) * a.B / a.NS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wave >> 1, mt = wave & 1;
    // taps of this block [p0, p1) split over the four tap groups; (part 0, group 0) also owns the bias
    const int p0 = tp * T / NP, np = (tp + 1) * T / NP - p0;
    const int t_begin = p0 + tg * np / 4;
    const int ntap = p0 + (tg + 1) * np / 4 - t_begin;
    const bool do_bias = (tp == 0 && tg == 0);     // wave-uniform
    int toff[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int t = min(t_begin + j, T - 1);
        toff[j] = (t / KS) * XRS + (t % KS) * 64;
    }

    for (int i = tid; i < XROWS * XRS / 16; i += 512)
        *reinterpret_cast<float4*>(ximg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < NPIX * GPS / 16; i += 512)
        *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    // lane roles in the transposed reads
    const int grp = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int p_lane = (grp >> 1) * 8 + q;                          // pixel of this lane within a k-step (+4h)
    const int ch_off = ((grp & 1) * 16 + 4 * pq) * 2;                // byte offset of its 4 channels
    const int g_lane = mt * 64 + ch_off;

    // staging roles (global -> registers -> LDS)
    float xv[2][8]; float gv[8]; int qc[8];
    auto prefetch = [&](int b) {
        const float* a1 = a.a1 + c * a.a1_cs + (size_t)b * CIN * CELLS1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int task = tid + 512 * u;
            if (task < CELLS1 * 4) {
                const float* src = a1 + (task & 3) * 8 * CELLS1 + (task >> 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[u][i] = src[i * CELLS1];
            }
        }
        if (tid < CELLS2 * 8) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const float* gs = a.g + c * a.g_cs + ((size_t)b * COUT + cg * 8) * CELLS2 + cell;
            const uint8_t* cs = a.code + c * a.c_cs + ((size_t)b * COUT + cg * 8) * CELLS2 + cell;
#pragma unroll
            for (int i = 0; i < 8; ++i) { gv[i] = gs[i * CELLS2]; qc[i] = cs[i * CELLS2]; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int task = tid + 512 * u;
            if (task < CELLS1 * 4) {
                const int p = task >> 2;
                *reinterpret_cast<bf16x8*>(ximg + (p / P1 + G::PAD) * XRS + (p % P1 + G::PAD) * 64 + (task & 3) * 16) =
                    pack_bf16x8(xv[u]);
            }
        }
        if (tid < CELLS2 * 8) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const int p00 = 2 * (cell / P2) * P1 + 2 * (cell % P2);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (qc[i] == qq) ? gv[i] : 0.f;
                *reinterpret_cast<bf16x8*>(gimg + (p00 + (qq >> 1) * P1 + (qq & 1)) * GPS + cg * 16) = pack_bf16x8(v);
            }
        }
    };

    f32x16 acc[NTAP], accb;
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NTAP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    };
    zero_acc();
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    if (b_begin < b_end) prefetch(b_begin);
    for (int item = b_begin; item < b_end; ++item) {
        __syncthreads();            // everyone finished reading the previous images (or zero-filling)
        commit();
        __syncthreads();
        if (item + 1 < b_end) prefetch(item + 1);

#pragma unroll 1
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int pa = ks * 16 + p_lane, pb = pa + 4;
            const unsigned char* ga = gimg + pa * GPS + g_lane;
            const bf16x8 af = tr_frag(ga, ga + 4 * GPS);
            const unsigned char* xa = ximg + (pa / P1) * XRS + (pa % P1) * 64 + ch_off;
            const unsigned char* xb = ximg + (pb / P1) * XRS + (pb % P1) * 64 + ch_off;
#pragma unroll
            for (int j = 0; j < NTAP; ++j) {
                if (j < ntap) {
                    const bf16x8 bfr = tr_frag(xa + toff[j], xb + toff[j]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
                }
            }
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, accb, 0, 0, 0);
        }
    }

    const int ci = lane & 31;
    if (a.part != nullptr) {
        // partial of this batch split, tap-major [t][n][ci] (a store instruction covers two 128-B runs; the
        // dW layout costs 64 scattered dwords per instruction: 80 vs 57 us for 10 clients), then the bias
        float* pw = a.part + ((size_t)c * a.NS + split) * (T * COUT * CIN + COUT);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            if (j < ntap) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pw[((t_begin + j) * COUT + mt * 32 + mfma_row_(e, lane)) * CIN + ci] = acc[j][e];
            }
        }
        if (do_bias && ci == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) pw[T * COUT * CIN + mt * 32 + mfma_row_(e, lane)] = accb[e];
        }
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
    if (do_bias && ci == 0) {
        float* db = a.db + c * a.db_cs;
#pragma unroll
        for (int e = 0; e < 16; ++e) db[mt * 32 + mfma_row_(e, lane)] = accb[e];
    }
}

// Weight gradient with NS-term split operands (PREC_F32_SPLIT). Three term images of both operands do not fit
// beside each other for all 64 gradient channels, so a block owns (client, 32-channel row tile, batch split):
// X terms 3 x 21 KB + G terms 3 x 13 KB = 103 KB, one 512-thread block per CU. Its 8 waves are 8 tap groups
// (25 taps: 3,3,3,3,3,3,3,4; the bias rides on wave 0); the k loop is instantiated per tap count so the MFMA
// stream is branch-free. a.NS counts the batch splits (a.NS > 1: partials + conv2_wgrad_reduce_kernel).
template <int KS, int NS>
__global__ __launch_bounds__(512, 1) void conv2_wgrad_split_kernel(WgradArgs a) {
    using G = Geo<KS>;
    constexpr int T = G::T;
    constexpr int NTAP = (T + 7) / 8;                    // most taps per wave
    constexpr int XRS = G::IW * 64 + ((((P1 - G::IW) * 64) % 256) + 256) % 256;
    constexpr int XROWS = G::IW + 1;                     // +1 zero row: K padding (pixels 196..207) reads it
    constexpr int GPS = 64;                              // the 32 channels of this block's row tile
    constexpr int KSTEPS = (CELLS1 + 15) / 16;           // 13
    constexpr int NPIX = KSTEPS * 16;                    // 208 (last 12 stay zero)
    constexpr int XIMG = XROWS * XRS, GIMG = NPIX * GPS;
    __shared__ __attribute__((aligned(16))) unsigned char ximg[NS * XIMG];
    __shared__ __attribute__((aligned(16))) unsigned char gimg[NS * GIMG];

    int c, sub;
    if (!xcd_map(2 * a.NS, a.K, c, sub)) return;
    const int mt = sub & 1, split = sub >> 1;
    const int b_begin = split * a.B / a.NS, b_end = (split + 1) * a.B / a.NS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int tg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t_begin = tg * T / 8;
    const int ntap = (tg + 1) * T / 8 - t_begin;        // NTAP or NTAP - 1
    const bool do_bias = tg == 0;                        // wave-uniform
    int toff[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int t = min(t_begin + j, T - 1);
        toff[j] = (t / KS) * XRS + (t % KS) * 64;
    }

    for (int i = tid; i < NS * XIMG / 16; i += 512)
        *reinterpret_cast<float4*>(ximg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < NS * GIMG / 16; i += 512)
        *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    // lane roles in the transposed reads
    const int grp = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int p_lane = (grp >> 1) * 8 + q;                          // pixel of this lane within a k-step (+4h)
    const int ch_off = ((grp & 1) * 16 + 4 * pq) * 2;                // byte offset of its 4 channels

    // staging roles (global -> registers -> LDS)
    float xv[2][8]; float gv[8]; int qc[8];
    auto prefetch = [&](int b) {
        const float* a1 = a.a1 + c * a.a1_cs + (size_t)b * CIN * CELLS1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int task = tid + 512 * u;
            if (task < CELLS1 * 4) {
                const float* src = a1 + (task & 3) * 8 * CELLS1 + (task >> 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[u][i] = src[i * CELLS1];
            }
        }
        if (tid < CELLS2 * 4) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const float* gs = a.g + c * a.g_cs + ((size_t)b * COUT + mt * 32 + cg * 8) * CELLS2 + cell;
            const uint8_t* cs = a.code + c * a.c_cs + ((size_t)b * COUT + mt * 32 + cg * 8) * CELLS2 + cell;
#pragma unroll
            for (int i = 0; i < 8; ++i) { gv[i] = gs[i * CELLS2]; qc[i] = cs[i * CELLS2]; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int task = tid + 512 * u;
            if (task < CELLS1 * 4) {
                const int p = task >> 2;
                bf16x8 tv[NS];
                split_bf16x8<NS>(xv[u], tv);
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    *reinterpret_cast<bf16x8*>(ximg + e * XIMG + (p / P1 + G::PAD) * XRS + (p % P1 + G::PAD) * 64 +
                                               (task & 3) * 16) = tv[e];
            }
        }
        if (tid < CELLS2 * 4) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const int p00 = 2 * (cell / P2) * P1 + 2 * (cell % P2);
            bf16x8 tv[NS];
            split_bf16x8<NS>(gv, tv);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    bf16x8 v;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (qc[i] == qq) ? tv[e][i] : (__bf16)0.f;
                    *reinterpret_cast<bf16x8*>(gimg + e * GIMG + (p00 + (qq >> 1) * P1 + (qq & 1)) * GPS + cg * 16) = v;
                }
            }
        }
    };

    f32x16 acc[NTAP], accb;
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // one sample's k loop with a compile-time tap count / bias flag
    auto kloop = [&](auto nt_c, auto bias_c) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll 1
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int pa = ks * 16 + p_lane, pb = pa + 4;
            const unsigned char* ga = gimg + pa * GPS + ch_off;
            const unsigned char* xa = ximg + (pa / P1) * XRS + (pa % P1) * 64 + ch_off;
            const unsigned char* xb = ximg + (pb / P1) * XRS + (pb % P1) * 64 + ch_off;
            bf16x8 af[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) af[e] = tr_frag(ga + e * GIMG, ga + e * GIMG + 4 * GPS);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bfr[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) bfr[e] = tr_frag(xa + e * XIMG + toff[j], xb + e * XIMG + toff[j]);
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bfr[eb], acc[j], 0, 0, 0);
            }
            if (BIAS) {
#pragma unroll
                for (int e = 0; e < NS; ++e) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[e], ones, accb, 0, 0, 0);
            }
        }
    };
    using std::integral_constant;

    if (b_begin < b_end) prefetch(b_begin);
    for (int item = b_begin; item < b_end; ++item) {
        __syncthreads();            // everyone finished reading the previous images (or zero-filling)
        commit();
        __syncthreads();
        if (item + 1 < b_end) prefetch(item + 1);
        if (do_bias) kloop(integral_constant<int, NTAP - 1>{}, integral_constant<bool, true>{});
        else if (ntap == NTAP) kloop(integral_constant<int, NTAP>{}, integral_constant<bool, false>{});
        else kloop(integral_constant<int, NTAP - 1>{}, integral_constant<bool, false>{});
    }

    const int ci = lane & 31;
    if (a.part != nullptr) {
        float* pw = a.part + ((size_t)c * a.NS + split) * (T * COUT * CIN + COUT);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            if (j < ntap) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pw[((t_begin + j) * COUT + mt * 32 + mfma_row_(e, lane)) * CIN + ci] = acc[j][e];
            }
        }
        if (do_bias && ci == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) pw[T * COUT * CIN + mt * 32 + mfma_row_(e, lane)] = accb[e];
        }
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
    if (do_bias && ci == 0) {
        float* db = a.db + c * a.db_cs;
#pragma unroll
        for (int e = 0; e < 16; ++e) db[mt * 32 + mfma_row_(e, lane)] = accb[e];
    }
}

// The same with ALL 64 gradient channels in one block (X terms 3 x 21 KB + G terms 3 x 26 KB = 146 KB): the input image
// is staged once per (client, sample) instead of once per row tile, a launch has K x NS blocks instead of 2 K x NS
// (50 clients: 5 batch splits = 250 blocks on 256 CUs; the two-tile form has 200 or 400), and the 2 x T (row tile, tap)
// units are dealt to the 8 waves as 7/6/6/6 + 6/7/6/6 (5x5; 3x3: 3/2/2/2 + 2/3/2/2) - waves w and w + 4 share a SIMD, so
// every SIMD carries 13 / 13 / 12 / 12 units and the two bias units ride on the light ones.
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv2_wgrad_split64_kernel(WgradArgs a) {
    constexpr int NS = 3;
    constexpr int NTHR = 64 * NW, TG = NW / 2;             // threads; tap groups per row tile
    using G = Geo<KS>;
    constexpr int T = G::T;
    constexpr int NTAP = (T + TG - 1) / TG;              // most taps per wave (T % TG == 1: one wave of a row tile has it)
    static_assert(T % TG == 1 && TG % 4 == 0, "tap split assumes 25 or 9 taps on 4 or 8 groups");
    constexpr int XRS = G::IW * 64 + ((((P1 - G::IW) * 64) % 256) + 256) % 256;
    constexpr int XROWS = G::IW + 1;                     // +1 zero row: K padding (pixels 196..207) reads it
    constexpr int GPS = 128;                             // all 64 gradient channels
    constexpr int KSTEPS = (CELLS1 + 15) / 16;           // 13
    constexpr int NPIX = KSTEPS * 16;                    // 208 (last 12 stay zero)
    constexpr int XIMG = XROWS * XRS, GIMG = NPIX * GPS;
    __shared__ __attribute__((aligned(16))) unsigned char ximg[NS * XIMG];
    __shared__ __attribute__((aligned(16))) unsigned char gimg[NS * GIMG];

    int c, split;
    if (!xcd_map(a.NS, a.K, c, split)) return;
    const int b_begin = split * a.B / a.NS, b_end = (split + 1) * a.B / a.NS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wv / TG;                              // row tile (32 gradient channels) of this wave
    const int jj = ((wv % TG) - mt) & (TG - 1);          // its tap group, rotated by the row tile (SIMD balance)
    const int t_begin = jj * (T / TG) + (jj > 0 ? 1 : 0);
    const int ntap = T / TG + (jj == 0 ? 1 : 0);        // NTAP or NTAP - 1
    const bool do_bias = jj == 2;                        // wave-uniform; a light wave on SIMD 2 (mt 0) / 3 (mt 1)
    int toff[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int t = min(t_begin + j, T - 1);
        toff[j] = (t / KS) * XRS + (t % KS) * 64;
    }

    for (int i = tid; i < NS * XIMG / 16; i += NTHR)
        *reinterpret_cast<float4*>(ximg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < NS * GIMG / 16; i += NTHR)
        *reinterpret_cast<float4*>(gimg + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    // lane roles in the transposed reads
    const int grp = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int p_lane = (grp >> 1) * 8 + q;                          // pixel of this lane within a k-step (+4h)
    const int ch_off = ((grp & 1) * 16 + 4 * pq) * 2;                // byte offset of its 4 channels (X; G: + 64 mt)

    // staging roles (global -> registers -> LDS)
    constexpr int XU = (CELLS1 * 4 + NTHR - 1) / NTHR;   // image staging tasks per thread: 2 / 1
    float xv[XU][8]; float gv[8]; int qc[8];
    auto prefetch = [&](int b) {
        const float* a1 = a.a1 + c * a.a1_cs + (size_t)b * CIN * CELLS1;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int task = tid + NTHR * u;
            if (task < CELLS1 * 4) {
                const float* src = a1 + (task & 3) * 8 * CELLS1 + (task >> 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[u][i] = src[i * CELLS1];
            }
        }
        if (tid < CELLS2 * 8) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const float* gs = a.g + c * a.g_cs + ((size_t)b * COUT + cg * 8) * CELLS2 + cell;
            const uint8_t* cs = a.code + c * a.c_cs + ((size_t)b * COUT + cg * 8) * CELLS2 + cell;
#pragma unroll
            for (int i = 0; i < 8; ++i) { gv[i] = gs[i * CELLS2]; qc[i] = cs[i * CELLS2]; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int task = tid + NTHR * u;
            if (task < CELLS1 * 4) {
                const int p = task >> 2;
                bf16x8 tv[NS];
                split_bf16x8<NS>(xv[u], tv);
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    *reinterpret_cast<bf16x8*>(ximg + e * XIMG + (p / P1 + G::PAD) * XRS + (p % P1 + G::PAD) * 64 +
                                               (task & 3) * 16) = tv[e];
            }
        }
        if (tid < CELLS2 * 8) {
            const int cg = tid / CELLS2, cell = tid % CELLS2;
            const int p00 = 2 * (cell / P2) * P1 + 2 * (cell % P2);
            bf16x8 tv[NS];
            split_bf16x8<NS>(gv, tv);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    bf16x8 v;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (qc[i] == qq) ? tv[e][i] : (__bf16)0.f;
                    *reinterpret_cast<bf16x8*>(gimg + e * GIMG + (p00 + (qq >> 1) * P1 + (qq & 1)) * GPS + cg * 16) = v;
                }
            }
        }
    };

    f32x16 acc[NTAP], accb;
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // one sample's k loop with a compile-time tap count / bias flag
    auto kloop = [&](auto nt_c, auto bias_c) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll 1
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int pa = ks * 16 + p_lane, pb = pa + 4;
            const unsigned char* ga = gimg + pa * GPS + mt * 64 + ch_off;
            const unsigned char* xa = ximg + (pa / P1) * XRS + (pa % P1) * 64 + ch_off;
            const unsigned char* xb = ximg + (pb / P1) * XRS + (pb % P1) * 64 + ch_off;
            bf16x8 af[NS];
#pragma unroll
            for (int e = 0; e < NS; ++e) af[e] = tr_frag(ga + e * GIMG, ga + e * GIMG + 4 * GPS);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bfr[NS];
#pragma unroll
                for (int e = 0; e < NS; ++e) bfr[e] = tr_frag(xa + e * XIMG + toff[j], xb + e * XIMG + toff[j]);
#pragma unroll
                for (int ea = 0; ea < NS; ++ea)
#pragma unroll
                    for (int eb = 0; eb < NS - ea; ++eb)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ea], bfr[eb], acc[j], 0, 0, 0);
            }
            if (BIAS) {
#pragma unroll
                for (int e = 0; e < NS; ++e) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[e], ones, accb, 0, 0, 0);
            }
        }
    };
    using std::integral_constant;

    if (b_begin < b_end) prefetch(b_begin);
    for (int item = b_begin; item < b_end; ++item) {
        __syncthreads();            // everyone finished reading the previous images (or zero-filling)
        commit();
        __syncthreads();
        if (item + 1 < b_end) prefetch(item + 1);
        if (do_bias) kloop(integral_constant<int, NTAP - 1>{}, integral_constant<bool, true>{});
        else if (ntap == NTAP) kloop(integral_constant<int, NTAP>{}, integral_constant<bool, false>{});
        else kloop(integral_constant<int, NTAP - 1>{}, integral_constant<bool, false>{});
    }

    const int ci = lane & 31;
    if (a.part != nullptr) {
        float* pw = a.part + ((size_t)c * a.NS + split) * (T * COUT * CIN + COUT);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            if (j < ntap) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pw[((t_begin + j) * COUT + mt * 32 + mfma_row_(e, lane)) * CIN + ci] = acc[j][e];
            }
        }
        if (do_bias && ci == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) pw[T * COUT * CIN + mt * 32 + mfma_row_(e, lane)] = accb[e];
        }
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
    if (do_bias && ci == 0) {
        float* db = a.db + c * a.db_cs;
#pragma unroll
        for (int e = 0; e < 16; ++e) db[mt * 32 + mfma_row_(e, lane)] = accb[e];
    }
}

// Sum of the NS batch-split partials [t][n][ci] -> dW[n][ci][t], db[n]. One block per (client, n): reads
// are 128-B runs, the transposition goes through LDS, the CIN*T floats of dW[n] leave as one contiguous run.
template <int KS>
__global__ __launch_bounds__(256) void conv2_wgrad_reduce_kernel(WgradArgs a) {
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

}  // namespace

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
constexpr int SPLIT_TERMS = 3;     // bf16 terms per operand of PREC_F32_SPLIT

// wp: per client NS * 2 * KS*KS * 64 * 32 bf16 (per term: forward copy, then mirrored data-gradient copy), NS = 1 for
// PREC_BF16 and 3 for PREC_F32_SPLIT
// (prec == PREC_F32: floats instead of bf16, forward copy [t][ci][co] + data-gradient copy [t'][co][ci]: conv2_f32.hip)
P2PFL_API int p2pfl_conv2_prep(const float* w, long w_cs, void* wp, long wp_cs, int K, int CIN_, int COUT_,
                               int KS, int prec, hipStream_t stream) {
    if (K <= 0 || CIN_ != CIN || COUT_ != COUT) return -1;
    if (prec == PREC_F32) return conv2_prep_f32(w, w_cs, (float*)wp, wp_cs, K, KS, stream);
    const int ns = prec == PREC_F32_SPLIT ? SPLIT_TERMS : 1;
    if (wp_cs < 2L * ns * KS * KS * COUT * CIN || (wp_cs & 7) || ((uintptr_t)wp & 15)) return -2;
    dim3 grid(cdiv(COUT * CIN * KS * KS, 256 * 4), K);
    __bf16* o = (__bf16*)wp;
    if (KS == 5 && ns == 1) hipLaunchKernelGGL((conv2_prep_kernel<5, 1>), grid, dim3(256), 0, stream, w, w_cs, o, wp_cs);
    else if (KS == 3 && ns == 1) hipLaunchKernelGGL((conv2_prep_kernel<3, 1>), grid, dim3(256), 0, stream, w, w_cs, o, wp_cs);
    else if (KS == 5) hipLaunchKernelGGL((conv2_prep_kernel<5, SPLIT_TERMS>), grid, dim3(256), 0, stream, w, w_cs, o, wp_cs);
    else if (KS == 3) hipLaunchKernelGGL((conv2_prep_kernel<3, SPLIT_TERMS>), grid, dim3(256), 0, stream, w, w_cs, o, wp_cs);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

// P2PFL_CONV2_DENSE=0: the window-row forward kernel for the exact-split policy too (A/B runs)
static bool fwd_dense() {
    static const bool on = [] { const char* e = getenv("P2PFL_CONV2_DENSE"); return !(e && atoi(e) == 0); }();
    return on;
}

// P2PFL_CONV2_DENSE_DGRAD=0: the 8-tile data-gradient kernel for the exact-split policy (A/B runs)
static bool dgrad_dense() {
    static const bool on = [] { const char* e = getenv("P2PFL_CONV2_DENSE_DGRAD"); return fwd_dense() && !(e && atoi(e) == 0); }();
    return on;
}

// samples a block of the dense forward walks through (the next sample's global loads fly under the tap loop of the
// current one); P2PFL_CONV2_IPB forces a count. ONE is best everywhere (tools/probe/c2_ipb.sh; 1 / 2 / 4 / 8 samples per
// block: 1000 clients 3x3 1268 / 1295 / 1301 / 1337 us, 50 clients 5x5 179 / 206 / 207 / 253, 10 clients 55 / 68 / 121 / 232):
// the kernel does not wait for its global loads - per tap its eight waves read 240 KB of fragments from LDS, 1920 cycles of
// the LDS pipe against 2496 of the matrix pipe - and more, shorter blocks balance better.
static int fwd_images_per_block(int K, int B) {
    static const int forced = [] { const char* e = getenv("P2PFL_CONV2_IPB"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced > B ? B : forced;
    return 1;
}

P2PFL_API int p2pfl_conv2_fwd(const float* a1, long a1_cs, const void* wp, long wp_cs, const float* bias,
                              long b_cs, float* a2, long a2_cs, uint8_t* code, long c_cs, int K, int B,
                              int CIN_, int COUT_, int KS, int prec, int flags, hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN_ != CIN || COUT_ != COUT) return -1;
    if (prec == PREC_F32)
        return conv2_fwd_f32(a1, a1_cs, (const float*)wp, wp_cs, bias, b_cs, a2, a2_cs, code, c_cs, K, B, KS, flags,
                             stream);
    const bool split = prec == PREC_F32_SPLIT;
    if ((wp_cs & 7) || ((uintptr_t)wp & 15) || wp_cs < 2L * (split ? SPLIT_TERMS : 1) * KS * KS * COUT * CIN) return -2;
    if (((uintptr_t)a2 & 15) || (a2_cs & 3) || ((uintptr_t)code & 15) || (c_cs & 15)) return -2;   // 16-byte block stores
    FwdArgs a{a1, a1_cs, (const __bf16*)wp, wp_cs, bias, b_cs, a2, a2_cs, code, c_cs, K, B, 1};
    dim3 grid(cdiv(K, 8) * 8 * (split ? B : cdiv(B, 2)));
    if (split && fwd_dense()) {
        a.ipb = fwd_images_per_block(K, B);
        grid = dim3(cdiv(K, 8) * 8 * cdiv(B, a.ipb));
    }
    if (KS == 5 && !split) hipLaunchKernelGGL((conv2_fwd_kernel<5, 1>), grid, dim3(256), 0, stream, a);
    else if (KS == 3 && !split) hipLaunchKernelGGL((conv2_fwd_kernel<3, 1>), grid, dim3(256), 0, stream, a);
    else if (KS == 5 && fwd_dense()) hipLaunchKernelGGL(conv2_fwd_dense_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 3 && fwd_dense()) hipLaunchKernelGGL(conv2_fwd_dense_kernel<3>, grid, dim3(256), 0, stream, a);
    else if (KS == 5) hipLaunchKernelGGL((conv2_fwd_kernel<5, SPLIT_TERMS>), grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL((conv2_fwd_kernel<3, SPLIT_TERMS>), grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_conv2_dgrad(const float* g, long g_cs, const uint8_t* code, long c_cs, const void* wp,
                                long wp_cs, float* da1, long da1_cs, int K, int B, int CIN_, int COUT_,
                                int KS, int prec, int flags, hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN_ != CIN || COUT_ != COUT) return -1;
    if (prec == PREC_F32)
        return conv2_dgrad_f32(g, g_cs, code, c_cs, (const float*)wp, wp_cs, da1, da1_cs, K, B, KS, flags, stream);
    const bool split = prec == PREC_F32_SPLIT;
    if ((wp_cs & 7) || ((uintptr_t)wp & 15) || (da1_cs & 3) || ((uintptr_t)da1 & 15)) return -2;
    if (wp_cs < 2L * (split ? SPLIT_TERMS : 1) * KS * KS * COUT * CIN) return -2;
    DgradArgs a{g, g_cs, code, c_cs, (const __bf16*)wp, wp_cs, da1, da1_cs, K, B};
    dim3 grid(cdiv(K, 8) * 8 * B);
    if (KS == 5 && !split) hipLaunchKernelGGL(conv2_dgrad_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 3 && !split) hipLaunchKernelGGL(conv2_dgrad_kernel<3>, grid, dim3(256), 0, stream, a);
    else if (KS == 5 && dgrad_dense()) hipLaunchKernelGGL(conv2_dgrad_dense_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 3 && dgrad_dense()) hipLaunchKernelGGL(conv2_dgrad_dense_kernel<3>, grid, dim3(256), 0, stream, a);
    else if (KS == 5) hipLaunchKernelGGL((conv2_dgrad_split_kernel<5, SPLIT_TERMS>), grid, dim3(256), 0, stream, a);
    else if (KS == 3) hipLaunchKernelGGL((conv2_dgrad_split_kernel<3, SPLIT_TERMS>), grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

static int num_cus() {
    static int n = 0;
    if (n <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
    }
    return n;
}

template <int KS>
static void launch_wgrad(const WgradArgs& a, int np, hipStream_t stream) {
    const dim3 block(512);
    const int slots = cdiv(a.K, 8) * 8 * a.NS;
    if (np == 1) hipLaunchKernelGGL((conv2_wgrad_kernel<KS, 1>), dim3(slots), block, 0, stream, a);
    else if (np == 2) hipLaunchKernelGGL((conv2_wgrad_kernel<KS, 2>), dim3(slots * 2), block, 0, stream, a);
    else hipLaunchKernelGGL((conv2_wgrad_kernel<KS, 5>), dim3(slots * 5), block, 0, stream, a);
    if (a.part != nullptr)
        hipLaunchKernelGGL(conv2_wgrad_reduce_kernel<KS>, dim3(COUT, a.K), dim3(256), 0, stream, a);
}

// Work split of the weight gradient for K clients: np tap parts x ns batch splits per client. Every block
// stages whole images however few taps it owns, so tap parts multiply the staging work while batch splits
// only add the partial-sum traffic: take the fewest tap parts with which <= 8 batch splits still give every
// CU a block (aiming at two).  3x3: 1000 clients -> 1 x 1, 50 -> 1 x 8;  5x5: 100 -> 2 x 3, 10 -> 5 x 8.
// 5x5 with one tap part means 7 accumulator tiles per wave: such a block does not fit on a CU beside the
// two workgroups of the fused wgrad+Adam kernel it is meant to overlap with (measured: it then waits the
// whole 3 ms out, headline 237 -> 255 ms), so below one client per CU the 5x5 kernel takes >= 2 parts.
static void wgrad_split(int K, int B, int KS, int cus, int& np, int& ns) {
    static const int parts[3] = {1, 2, 5};
    for (int i = (KS == 5 && K < cus) ? 1 : 0; i < 3; ++i) {
        np = parts[i];
        ns = (2 * cus + K * np - 1) / (K * np);
        ns = ns > 8 ? 8 : ns;
        ns = ns > B ? B : ns;
        ns = ns < 1 ? 1 : ns;
        if ((long)K * np * ns >= cus) return;
    }
}

// Which filter sizes take the 64-channel blocks of conv2_wgrad_split64_kernel: 3x3 (1000 clients 1741 -> 1421 us, 50 clients
// 124 -> 112). 5x5 does not pay: with 8 waves a wave has 7 accumulator tiles beside the staging registers and spills (50
// clients: 250 us against 225-230 for the two-row-tile form), with 16 waves (4 tiles per wave, the instance kept) the
// 128-register file of a wave spills as well (255 us). P2PFL_WGRAD_WIDE=0 / 1: none / both (A/B runs).
static bool wgrad_wide_for(int KS) {
    static const int mode = [] { const char* e = getenv("P2PFL_WGRAD_WIDE"); return e ? atoi(e) : -1; }();
    return mode < 0 ? KS == 3 : mode != 0;
}

// Batch splits of the split-operand weight gradient (one block per CU at a time, 2 * K * ns blocks): the count
// <= 8 with the best (CU fill of the last round) x (evenness of the sample split). More splits must buy > 15 %:
// partials cost a second kernel (measured, 100 clients x 32 samples: 1 split 0.426 ms, 5 splits 0.441 ms although
// the blocks alone are 14 % better packed; 16 splits 0.548 ms). 100 clients -> 1; 1000 -> 1; 10 -> 8.
static int wgrad_split_batches(int K, int B, int cus, int KS) {
    static const int forced = [] { const char* e = getenv("P2PFL_WGRAD_NS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced > B ? B : forced;
    int best = 1;
    double best_eff = 0.0;
    if (wgrad_wide_for(KS)) {
        // 64-channel blocks: K * ns of them, the blocks of a client on ONE XCD (xcd_map) - what has to come out even is the
        // number of block rounds of the fullest XCD (50 clients: 7 on an XCD; 4 splits = 28 blocks on its 32 CUs 112 us,
        // 5 splits = 35 blocks 161 us, 8 splits = 56 blocks 116 us; tools/probe/c2w_wide.sh)
        const int per_xcd = cus / 8 > 0 ? cus / 8 : 1;
        for (int ns = 1; ns <= 8 && ns <= B; ++ns) {
            const int rounds = cdiv(cdiv(K, 8) * ns, per_xcd);
            const double fill = (double)K * ns / (8.0 * rounds * per_xcd);
            const double even = ((double)B / ns) / (double)cdiv(B, ns);
            if (fill * even > best_eff * (best_eff > 0.0 ? 1.15 : 1.0) + 1e-9) { best_eff = fill * even; best = ns; }
        }
        return best;
    }
    for (int ns = 1; ns <= 8 && ns <= B; ++ns) {
        const long blocks = 2L * K * ns;
        const double fill = (double)blocks / (double)(cdiv((int)blocks, cus) * (long)cus);
        const double even = ((double)B / ns) / (double)cdiv(B, ns);
        if (fill * even > best_eff * (best_eff > 0.0 ? 1.15 : 1.0) + 1e-9) { best_eff = fill * even; best = ns; }
    }
    return best;
}

// floats of scratch p2pfl_conv2_wgrad wants for this problem (0: none)
P2PFL_API long p2pfl_conv2_wgrad_scratch(int K, int B, int KS, int prec) {
    if (K <= 0 || B <= 0) return 0;
    if (prec == PREC_F32) return conv2_wgrad_f32_scratch(K, B, KS);
    int np, ns;
    if (prec == PREC_F32_SPLIT) ns = wgrad_split_batches(K, B, num_cus(), KS);
    else wgrad_split(K, B, KS, num_cus(), np, ns);
    return ns > 1 ? (long)K * ns * (KS * KS * COUT * CIN + COUT) : 0;
}

// ``opt`` (optional): the optimizer step of this layer runs in the reduction pass that holds the final gradient - dw / db
// are then NOT written (may be null) and the partial-sum path is taken even for one batch split, so ``scratch`` must
// hold K * max(ns, 1) * (KS*KS*64*32 + 64) floats (-5 otherwise).
P2PFL_API int p2pfl_conv2_wgrad_opt(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* a1,
                                    long a1_cs, float* dw, long dw_cs, float* db, long db_cs, float* scratch,
                                    long scratch_floats, int K, int B, int CIN_, int COUT_, int KS, int prec,
                                    int flags, const P2pflOpt* opt, hipStream_t stream) {
    if (K <= 0 || B <= 0 || CIN_ != CIN || COUT_ != COUT) return -1;
    if (!opt_epi_valid(opt)) return -1;
    const bool fold = opt != nullptr && opt->kind != 0;
    if (prec == PREC_F32)
        return conv2_wgrad_f32(g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, scratch, scratch_floats, K, B, KS,
                               flags, opt, stream);
    if (KS != 5 && KS != 3) return -3;
    const long pw = (long)KS * KS * COUT * CIN + COUT;
    if (prec == PREC_F32_SPLIT) {
        int ns = wgrad_split_batches(K, B, num_cus(), KS);
        if (scratch == nullptr || scratch_floats < (long)K * ns * pw) ns = 1;
        if (fold && (scratch == nullptr || scratch_floats < (long)K * ns * pw)) return -5;
        WgradArgs a{g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, K, B, (ns > 1 || fold) ? scratch : nullptr, ns,
                    make_opt_epi(opt)};
        if (wgrad_wide_for(KS)) {
            const dim3 grid(cdiv(K, 8) * 8 * ns);
            if (KS == 5) hipLaunchKernelGGL((conv2_wgrad_split64_kernel<5, 16>), grid, dim3(1024), 0, stream, a);
            else hipLaunchKernelGGL((conv2_wgrad_split64_kernel<3, 8>), grid, dim3(512), 0, stream, a);
        } else {
        const dim3 grid(cdiv(K, 8) * 8 * 2 * ns);
        if (KS == 5) hipLaunchKernelGGL((conv2_wgrad_split_kernel<5, SPLIT_TERMS>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((conv2_wgrad_split_kernel<3, SPLIT_TERMS>), grid, dim3(512), 0, stream, a);
        }
        if (a.part != nullptr) {
            if (KS == 5) hipLaunchKernelGGL(conv2_wgrad_reduce_kernel<5>, dim3(COUT, K), dim3(256), 0, stream, a);
            else hipLaunchKernelGGL(conv2_wgrad_reduce_kernel<3>, dim3(COUT, K), dim3(256), 0, stream, a);
        }
        P2PFL_RETURN_LAST_ERROR();
    }
    int np, ns;
    wgrad_split(K, B, KS, num_cus(), np, ns);
    if (scratch == nullptr || scratch_floats < (long)K * ns * pw) ns = 1;
    if (fold && (scratch == nullptr || scratch_floats < (long)K * ns * pw)) return -5;
    WgradArgs a{g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, K, B, (ns > 1 || fold) ? scratch : nullptr, ns,
                make_opt_epi(opt)};
    if (KS == 5) launch_wgrad<5>(a, np, stream);
    else launch_wgrad<3>(a, np, stream);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_conv2_wgrad(const float* g, long g_cs, const uint8_t* code, long c_cs, const float* a1,
                                long a1_cs, float* dw, long dw_cs, float* db, long db_cs, float* scratch,
                                long scratch_floats, int K, int B, int CIN_, int COUT_, int KS, int prec,
                                int flags, hipStream_t stream) {
    return p2pfl_conv2_wgrad_opt(g, g_cs, code, c_cs, a1, a1_cs, dw, dw_cs, db, db_cs, scratch, scratch_floats, K, B,
                                 CIN_, COUT_, KS, prec, flags, nullptr, stream);
}
