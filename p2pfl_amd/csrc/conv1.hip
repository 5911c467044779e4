// Stage-1 convolution (C_in = 1; 5x5 or 7x7, "same" padding, + ReLU + 2x2 max-pool) of the MNIST / FEMNIST
// CNNs on the gfx950 matrix cores, client-batched.
//
// K = 25 / 49 taps is thin for a GEMM, but with the taps laid out as k = ky*8 + kx (kx padded to 8) an
// MFMA operand fragment (8 consecutive k) is simply 8 consecutive pixels of one image row: the padded
// image of a sample sits in LDS as bf16 and every fragment is ONE ds_read_b128 at a pixel address
// (gfx950 LDS reads need no alignment), no im2col, no gather.
//
//   forward   C_q[f][window] = sum_k W[f][k] X_q[window][k]   one wave = one image, 7 tiles of 32 pool windows, one
//             accumulator per window position q: a lane ends up with the four pixels of ITS window for 16
//             filters, so max / arg-max / ReLU are in-lane compares and every store instruction writes 32
//             consecutive pooled outputs (128 B) of a filter. (A first version that put the four pixels of
//             a window into four lanes spent 8x the VALU work on DPP exchanges and stored 32-byte pieces:
//             0.93 ms vs 1.2 ms for the VALU kernel it replaced, 1000 FEMNIST clients.)
//   weight    dW[f][k] = sum_{q, window} G_q[f][window] X_q[window][k]     the pooled gradient is routed to the
//   gradient  arg-max pixel by masking with the code (4 masks of ONE operand read); the image is kept
//             de-interleaved by column parity so that 8 consecutive windows of a window row are 8 consecutive
//             bf16; tap 63 is an all-ones column whose product is the bias gradient. A block streams the
//             images of a client through double-buffered LDS with coalesced loads (gradient, codes, pixels);
//             its four waves split the 14 window rows; fp32 atomics of the per-wave [32 x 64] partials.
//
// Reference layers: fedstellar/learning/pytorch/mnist/models/cnn.py:134-138,151-152 (5x5),
// fedstellar/learning/pytorch/femnist/models/cnn.py:136,147 (7x7).
#include "conv_f32.h"

namespace {

constexpr int IMG = 28;
constexpr int P1 = 14;
constexpr int CELLS1 = 196;

__device__ __forceinline__ int mfma_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

__device__ __forceinline__ bf16x8 lds_read8(const __bf16* p) {
    bf16x8 v;
    __builtin_memcpy(&v, p, 16);   // ds_read_b128 at a 2-byte aligned address
    return v;
}

// A fragments of the filter bank: lane (f = lane & 31, h = lane >> 5), k-step kk covers taps ky = 2kk + h, kx = 0..7
template <int KS, int NKK>
__device__ __forceinline__ void load_filter_frags(const float* w, int F, bf16x8 (&wf)[NKK]) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int ky = 2 * kk + h;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            wf[kk][j] = (__bf16)((r < F && ky < KS && j < KS) ? w[r * KS * KS + ky * KS + j] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
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
__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1Args a) {
    constexpr int PAD = KS / 2;
    constexpr int NKK = (KS * 8 + 15) / 16;     // k-steps of 16 taps (two filter rows each): 3 / 4
    constexpr int ROWS = IMG + 2 * NKK;         // last pixel row 27 + last tap row 2 NKK - 1
    constexpr int LW = 36;                      // last pixel column 27 + kx 7 = 34
    constexpr int NPX = IMG * IMG / 4;          // float4 per image
    constexpr int OT_LD = 40, CT_LD = 48;       // output patch rows: 32 windows + pad (bank spread / 16-B rows)
    // one private image + output patch per wave: no workgroup barrier anywhere in this kernel
    __shared__ __attribute__((aligned(16))) __bf16 img[4][ROWS * LW + 8];
    __shared__ __attribute__((aligned(16))) float otile[4][32 * OT_LD];
    __shared__ __attribute__((aligned(16))) uint8_t ctile[4][32 * CT_LD];

    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    __bf16* my = img[wave];
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
    for (int i = lane; i < (ROWS * LW + 8) / 2; i += 64) reinterpret_cast<uint32_t*>(my)[i] = 0u;   // zero border, once
    bf16x8 wf[NKK];
    load_filter_frags<KS, NKK>(a.w + c * a.w_cs, a.F, wf);
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
                bf16x4 q;
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = (__bf16)px[i][j];
                __builtin_memcpy(my + (y + PAD) * LW + x + PAD, &q, 8);
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
                const __bf16* p = my + (2 * py + half + h) * LW + 2 * pxx;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], lds_read8(p + d + 2 * kk * LW), acc[d], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (half == 0) { best[e] = acc[0][e]; qi[e] = 0; }
                    else if (acc[0][e] > best[e]) { best[e] = acc[0][e]; qi[e] = 2; }
                    if (acc[1][e] > best[e]) { best[e] = acc[1][e]; qi[e] = 2 * half + 1; }
                }
            }
            // bias + ReLU, then the tile (32 windows x 32 filters) goes through a per-wave LDS patch so that the
            // stores are 16 bytes per lane in address order: 5 store instructions per tile instead of 32
            // (measured: the 4-byte / 1-byte per lane stores cost as much as all the arithmetic).
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
                const int f = lane >> 1, half = lane & 1, w0 = tile * 32 + 16 * half;
                uint8_t* dst = code + (size_t)f * CELLS1 + w0;
                if (f < a.F && w0 + 16 <= CELLS1)
                    *reinterpret_cast<u32x4_a4*>(dst) = *reinterpret_cast<const u32x4_a4*>(ct + f * CT_LD + 16 * half);
                else if (f < a.F && w0 < CELLS1)          // last tile: windows 192..195
                    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(ct + f * CT_LD + 16 * half);
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
    float* part;                       // per-wave partial tiles [K][gridDim.x * 4][2][16][64] (raw accumulator layout)
};

constexpr int WG_ROWS = 36;     // last row 2*13 + 1 + tap row 7 = 34
constexpr int WG_HW = 24;       // half-row: last index 8 + 7 + (8 >> 1) = 19
constexpr int WG_IMG = WG_ROWS * WG_HW + 8;
constexpr int WG_G = 32 * CELLS1 + 8;

struct Conv1WgradSmem {
    __bf16 img[2][WG_IMG];      // [parity][row][i] = padded pixel (row, 2 i + parity)
    __bf16 g[WG_G];             // pooled gradient [f][196]
    uint8_t code[WG_G];         // arg-max codes   [f][196]
};

// registers of one image in flight (coalesced global loads; 256 threads)
template <int KS>
struct Conv1WgradStage {
    static constexpr int PAD = KS / 2;
    static constexpr int NG = (32 * CELLS1 / 4 + 255) / 256;   // float4 of the gradient per thread: 7
    static constexpr int NC = (32 * CELLS1 / 4 + 255) / 256;   // code words per thread: 7
    static constexpr int NP = (2 * WG_IMG + 255) / 256;        // staged pixels per thread: 7
    f32x4 g[NG]; uint32_t c[NC]; float px[NP];

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
            px[i] = (o2 < 2 * WG_IMG && o < WG_ROWS * WG_HW && y >= 0 && y < IMG && x >= 0 && x < IMG) ? src[y * IMG + x] : 0.f;
        }
    }
    __device__ __forceinline__ void store(const Conv1WgradArgs& a, Conv1WgradSmem& sm) const {
        const int n4 = a.F * CELLS1 / 4;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int o = threadIdx.x + i * 256;
            if (o < n4) {
                bf16x4 q;
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = (__bf16)g[i][j];
                *reinterpret_cast<bf16x4*>(sm.g + 4 * o) = q;
                *reinterpret_cast<uint32_t*>(sm.code + 4 * o) = c[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int o2 = threadIdx.x + i * 256;
            if (o2 < 2 * WG_IMG) (&sm.img[0][0])[o2] = (__bf16)px[i];
        }
    }
};

template <int KS>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(Conv1WgradArgs a) {
    __shared__ __attribute__((aligned(16))) Conv1WgradSmem smem[2];

    const int c = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    // rows of filters >= F (and the read-ahead pad) stay "no gradient" for the whole kernel
    for (int i = threadIdx.x; i < 2 * WG_G; i += 256) {
        Conv1WgradSmem& s = smem[i / WG_G];
        const int o = i % WG_G;
        s.g[o] = (__bf16)0.f;
        s.code[o] = 4;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;

    Conv1WgradStage<KS> st;
    int b = blockIdx.x;
    if (b < a.B) { st.load(a, c, b); st.store(a, smem[0]); }
    __syncthreads();
    int cur = 0;
    for (; b < a.B; b += gridDim.x) {
        const bool has_next = b + (int)gridDim.x < a.B;
        if (has_next) st.load(a, c, b + gridDim.x);          // global loads fly while this image is consumed
        const Conv1WgradSmem& sm = smem[cur];
        for (int py = wave; py < P1; py += 4) {
            // pooled gradient + codes of 8 windows (px = 8h .. 8h+7; px >= 14 is padding) of filter r
            const int o = r * CELLS1 + py * P1 + 8 * h;
            const bf16x8 gv = lds_read8(sm.g + o);
            uint8_t cv[8];
            __builtin_memcpy(cv, sm.code + o, 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 af;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool on = cv[j] == q && (j < 6 || h == 0);
                    af[j] = on ? gv[j] : (__bf16)0.f;
                }
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    // B rows: taps n = jn*32 + r = ky*8 + kx; pixel of window px: (2py + dy + ky, 2px + dx + kx)
                    const int n = jn * 32 + r, ky = n >> 3, kx = n & 7;
                    const int v = (q & 1) + kx;
                    const __bf16* p = sm.img[v & 1] + (2 * py + (q >> 1) + ky) * WG_HW + 8 * h + (v >> 1);
                    const bf16x8 bfr = (n == 63) ? ones : lds_read8(p);
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[jn], 0, 0, 0);
                }
            }
        }
        if (has_next) st.store(a, smem[cur ^ 1]);
        __syncthreads();
        cur ^= 1;
    }
    // acc[jn][e] = partial dW[f = mfma_row(e)][tap n = jn*32 + r] of this wave's window rows: stored raw (coalesced, no
    // atomics); conv1_wgrad_finish_kernel adds the partials of a client up in a fixed order - bitwise reproducible
    float* pw = a.part + (((size_t)c * gridDim.x + blockIdx.x) * 4 + wave) * (2 * 1024);
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(jn * 16 + e) * 64 + lane] = acc[jn][e];
}

// Sum of the per-wave partial tiles of one client, partial by partial in index order, then either the gradient store or
// the layer's optimizer step (the gradient value is final here, so it never has to reach memory).
// A block owns a quarter of one 32 x 32 tile (jn) of one client, thread = (accumulator register e, lane) of the MFMA
// tile layout: 4 * njn blocks per client - one 1024-thread block per client left 10-client shards with 10 blocks, each
// walking up to 128 partials one dependent load after the other (87 us per step on the critical chain). The loads of up to 32
// partials are in flight together; the ADDS stay in index order, so the sum is bit-equal to the one-by-one loop.
struct Conv1FinishArgs {
    const float* part; int nparts, njn, tap8;
    float* dw; long dw_cs; float* db; long db_cs;
    int F, KS;
    OptEpi opt;
};

__global__ __launch_bounds__(256) void conv1_wgrad_finish_kernel(Conv1FinishArgs a) {
    const int c = blockIdx.x, jn = blockIdx.y >> 2;
    const int t = (blockIdx.y & 3) * 256 + threadIdx.x;           // index into the 1024-float tile
    const int lane = t & 63, e = t >> 6;
    const int f = mfma_row(e, lane), T = a.KS * a.KS;
    const OptEpi::Scal sc = a.opt.kind ? a.opt.scal() : OptEpi::Scal{0.f, 0.f};
    const float* p = a.part + ((size_t)c * a.nparts * a.njn + jn) * 1024 + t;
    const size_t ks = (size_t)a.njn * 1024;
    float s = 0.f;
    int k = 0;
    // (32 loads in flight: a 10-client shard has 128 partials per sum - 4 memory round trips instead of 16)
    for (; k + 32 <= a.nparts; k += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = p[(size_t)(k + u) * ks];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
    }
    for (; k + 8 <= a.nparts; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * ks];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < a.nparts; ++k) s += p[(size_t)k * ks];
    if (f >= a.F) return;
    const int n = jn * 32 + (lane & 31);
    int tap = -1; bool bias = false;
    if (a.tap8) {
        const int ky = n >> 3, kx = n & 7;
        if (n == 63) bias = true;
        else if (ky < a.KS && kx < a.KS) tap = ky * a.KS + kx;
    } else {
        if (n == T) bias = true;
        else if (n < T) tap = n;
    }
    if (bias) {
        if (a.opt.kind) a.opt.bias(sc, c, f, s);
        else a.db[c * a.db_cs + f] = s;
    } else if (tap >= 0) {
        if (a.opt.kind) a.opt.weight(sc, c, f * T + tap, s);
        else a.dw[c * a.dw_cs + f * T + tap] = s;
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

P2PFL_API int p2pfl_conv1_fwd(const float* data, long data_cs, const int* sidx, long sidx_cs,
                              const float* w, long w_cs, const float* bias, long b_cs, float* a1,
                              long a_cs, uint8_t* code, long c_cs, int K, int B, int F, int KS, int prec,
                              hipStream_t stream) {
    if (prec != PREC_BF16)
        return conv1_fwd_f32(data, data_cs, sidx, sidx_cs, w, w_cs, bias, b_cs, a1, a_cs, code, c_cs, K, B, F, KS, stream);
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    if (((uintptr_t)data & 15) || (data_cs & 3)) return -2;      // float4 loads of whole 784-pixel images
    if (((uintptr_t)a1 & 15) || (a_cs & 3) || ((uintptr_t)code & 3) || (c_cs & 3)) return -2;   // 16-byte output pieces
    // images per wave: amortise the filter / bias loads once there are >= ~2 waves per SIMD anyway
    int ipw = (int)(((long)K * B / 4) / 2048);
    if (ipw < 1) ipw = 1;
    if (ipw > 8) ipw = 8;
    Conv1Args a{data, data_cs, sidx, sidx_cs, w, w_cs, bias, b_cs, a1, a_cs, code, c_cs, B, F, ipw};
    dim3 grid(cdiv(B, 4 * ipw), K);
    if (KS == 5) hipLaunchKernelGGL(conv1_fwd_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL(conv1_fwd_kernel<7>, grid, dim3(256), 0, stream, a);
    else return -3;
    P2PFL_RETURN_LAST_ERROR();
}

int conv1_wgrad_finish(const float* part, int nparts, int njn, int tap8, float* dw, long dw_cs, float* db, long db_cs,
                       int K, int F, int KS, const P2pflOpt* opt, hipStream_t stream) {
    if (!opt_epi_valid(opt)) return -1;
    Conv1FinishArgs a{part, nparts, njn, tap8, dw, dw_cs, db, db_cs, F, KS, make_opt_epi(opt)};
    if (a.opt.kind == 0 && (dw == nullptr || db == nullptr)) return -1;
    if (K > 65535 || njn < 1) return -1;
    hipLaunchKernelGGL(conv1_wgrad_finish_kernel, dim3(K, 4 * njn), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

// blocks per client of the bf16 kernel: about 1.5 per CU over the launch, at most one per image
// measured (7x7; 10 / 50 / 100 clients): target 256: 49 / 63 / 107 us, 384: 63 / 63 / 91, 512: 63 / 67 / 87, 768: 63 / 88 / 110, 1536: 64 / 124 / 140
static int conv1_wgrad_splits(int K, int B) {
    int splits = cdiv(384, K);
    if (splits > B) splits = B;
    if (splits < 1) splits = 1;
    return splits;
}

// floats of scratch p2pfl_conv1_wgrad needs (the per-wave partial tiles)
P2PFL_API long p2pfl_conv1_wgrad_scratch(int K, int B, int KS, int prec) {
    if (K <= 0 || B <= 0) return 0;
    if (prec != PREC_BF16) return (long)K * conv1_wgrad_f32_splits(K, B) * 4 * ((KS * KS + 1 + 31) / 32) * 1024;
    return (long)K * conv1_wgrad_splits(K, B) * 4 * 2 * 1024;
}

// Stage-1 weight / bias gradient. Blocks write per-wave partial tiles into ``scratch`` and a finishing pass adds them up
// in a fixed order (no atomics: bitwise reproducible) and either stores dw / db or - ``opt`` given - applies the layer's
// optimizer step right there (dw / db may then be null).
P2PFL_API int p2pfl_conv1_wgrad(const float* g, long g_cs, const uint8_t* code, long c_cs,
                                const float* data, long data_cs, const int* sidx, long sidx_cs,
                                float* dw, long dw_cs, float* db, long db_cs, float* scratch, long scratch_floats,
                                int K, int B, int F, int KS, int prec, const P2pflOpt* opt, hipStream_t stream) {
    if (!opt_epi_valid(opt)) return -1;
    if (prec != PREC_BF16)
        return conv1_wgrad_f32(g, g_cs, code, c_cs, data, data_cs, sidx, sidx_cs, dw, dw_cs, db, db_cs, scratch,
                               scratch_floats, K, B, F, KS, opt, stream);
    if (K <= 0 || B <= 0 || F <= 0 || F > 32) return -1;
    // float4 / word loads of the pooled gradient and the codes of an image
    if (((uintptr_t)g & 15) || (g_cs & 3) || ((uintptr_t)code & 3) || (c_cs & 3)) return -2;
    const int splits = conv1_wgrad_splits(K, B);
    if (scratch == nullptr || scratch_floats < (long)K * splits * 4 * 2 * 1024) return -5;
    Conv1WgradArgs a{g, g_cs, code, c_cs, data, data_cs, sidx, sidx_cs, dw, dw_cs, db, db_cs, B, F, scratch};
    dim3 grid(splits, 1, K);
    if (KS == 5) hipLaunchKernelGGL(conv1_wgrad_kernel<5>, grid, dim3(256), 0, stream, a);
    else if (KS == 7) hipLaunchKernelGGL(conv1_wgrad_kernel<7>, grid, dim3(256), 0, stream, a);
    else return -3;
    return conv1_wgrad_finish(scratch, splits * 4, 2, /*tap8=*/1, dw, dw_cs, db, db_cs, K, F, KS, opt, stream);
}
