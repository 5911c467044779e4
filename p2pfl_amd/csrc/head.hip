// The classification head of a training step in ONE kernel per client:
//
//     logits = x W^T + b  ->  softmax cross-entropy (+ loss / accuracy / confusion counters)  ->  dlogits
//     dx = (dlogits W) * [x > 0]        (data gradient for the layer below, ReLU mask of ITS output = this x)
//     dW = dlogits^T x,  db = sum_b dlogits   ->  Adam / SGD-momentum step of (W, b) in place (or gradient store)
//
// i.e. nn.Linear(In, C) + CrossEntropyLoss + their backward + the optimizer step of the reference models
// (fedstellar/learning/pytorch/mnist/models/cnn.py:144,158,161-177; mnist/models/mlp.py:148-167). With C = 10 classes
// these are four launch-bound kernels of 5..50 us each (skinny forward, loss, data gradient, weight gradient) plus the
// element-wise optimizer on the dependent chain of every step - 102 times per headline round beside the HBM-bound
// update of the other client group, where each of them also waits for wave slots (141 us per group step in
// profiles/r2_final_round_timeline.txt). One 1024-thread workgroup per client does all of it: the 16 waves split the
// reduction of the logits on the matrix cores (fp32 MFMA), one barrier later every wave has the 32 x C gradient in LDS,
// and then a THREAD owns a column k of W: it holds W[:, k] (C registers) and x[:, k] (32 registers), so dx[:, k], dW[:, k]
// and the optimizer update of W[:, k] are register arithmetic with coalesced global accesses and no further barrier.
#include <cstdlib>

#include "gemm_core.h"
#include "opt_epi.h"

namespace {

struct HeadArgs {
    const float* x; long x_cs;        // [K][B][In]  input activations (post-ReLU output of the layer below)
    float* w;       long w_cs;        // [K][C][In]  weights (updated in place when opt.kind != 0: opt.pw == w)
    float* b;       long b_cs;        // [K][C]
    float* dx;      long dx_cs;       // [K][B][In]  (null: no data gradient wanted)
    float* dw;      long dw_cs;       // [K][C][In]  gradient outputs when opt.kind == 0
    float* db;      long db_cs;       // [K][C]
    float* logits;  long lo_cs;       // optional [K][B][C]
    const int* labels; long y_cs;     // [K][n_samples]
    const int* sidx;   long s_cs;     // [K][B] sample gather for the labels (or null)
    float* loss_sum; int* correct; int* count; int* confusion;
    const int* n_valid; long nv_cs;
    int B, In, C, relu_mask, vec;
    OptEpi opt;
};

constexpr int HEAD_CMAX = 16;         // most classes; a kernel instance pads them to HEAD_CP = 4, 8, 12 or 16 registers

template <int U, int HEAD_CP>
__global__ __launch_bounds__(1024) void linear_head_kernel(HeadArgs a) {
    __shared__ float red[16][16][64];                 // per-wave partial logit tiles
    __shared__ __attribute__((aligned(16))) float s_lg[32][HEAD_CMAX];
    __shared__ __attribute__((aligned(16))) float s_dl[32][HEAD_CMAX];
    __shared__ float s_loss[16];
    __shared__ int s_corr[16];

    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, kq = (lane >> 5) * 8;
    const float* x = a.x + c * a.x_cs;
    float* w = a.w + c * a.w_cs;
    float* bias = a.b + c * a.b_cs;
    const int B = a.B, In = a.In, C = a.C;

    // ---- phase 1: logits on the matrix cores, the 16 waves split the reduction (as linear_fwd_skinny_kernel) ----
    {
        LoadRowMajorK xl{x, B, In, In, (bool)(a.vec & 1)};
        LoadRowMajorK wl{w, C, In, In, (bool)(a.vec & 2)};
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int nsteps = (In + 15) / 16;
        for (int s0 = wave; s0 < nsteps; s0 += 16 * U) {
            float xv[U][8], wv[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = (s0 + 16 * u < nsteps) ? (s0 + 16 * u) * 16 + kq : In;     // past the end: zeros
                xl.load8(r, k, xv[u]);
                wl.load8(r, k, wv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u][e], wv[u][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[e];
    }
    __syncthreads();
    {
        const int e_out = threadIdx.x >> 6;            // thread (e, lane) owns logits[m][n]
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) v += red[wv][e_out][lane];
        const int m = mfma_row(e_out, lane), n = r;
        if (n < HEAD_CP) s_lg[m][n] = n < C ? v + bias[n] : 0.f;
    }
    __syncthreads();

    // ---- phase 2: softmax cross-entropy; wave w takes rows w and w + 16, a lane owns one class ----
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], B) : B;
    {
        const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
        float lsum = 0.f; int lcorrect = 0;           // wave-uniform
        for (int b = wave; b < 32; b += 16) {
            if (b >= nv) {
                if (lane < HEAD_CP) s_dl[b][lane] = 0.f;
                continue;
            }
            const int sample = a.sidx ? a.sidx[c * a.s_cs + b] : b;
            const int y = a.labels[c * a.y_cs + sample];
            const float v = lane < C ? s_lg[b][lane] : -INFINITY;
            if (a.logits && lane < C) a.logits[c * a.lo_cs + (size_t)b * C + lane] = v;
            float mx = v; int am = lane < C ? lane : 0x7fffffff;       // max and FIRST arg-max
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float om = __shfl_xor(mx, off, 64);
                const int oa = __shfl_xor(am, off, 64);
                if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
            }
            const float ex = lane < C ? expf(v - mx) : 0.f;
            const float se = wave_reduce_sum(ex);
            const float zy = __shfl(v, y, 64);
            lsum += mx + logf(se) - zy;
            lcorrect += (am == y);
            if (lane < HEAD_CP) s_dl[b][lane] = lane < C ? (ex / se - (lane == y ? 1.f : 0.f)) * inv : 0.f;
            if (a.confusion && lane == 0) atomicAdd(a.confusion + ((size_t)c * C + y) * C + am, 1);
        }
        if (lane == 0) { s_loss[wave] = lsum; s_corr[wave] = lcorrect; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f; int k = 0;
#pragma unroll
        for (int wv = 0; wv < 16; ++wv) { l += s_loss[wv]; k += s_corr[wv]; }
        a.loss_sum[c] += l;
        a.correct[c] += k;
        a.count[c] += nv;
    }

    // ---- phase 3: a thread owns column k of W: data gradient, weight gradient, optimizer step ----
    const OptEpi::Scal sc = a.opt.kind ? a.opt.scal() : OptEpi::Scal{0.f, 0.f};
    float* dx = a.dx ? a.dx + c * a.dx_cs : nullptr;
    for (int k = threadIdx.x; k < In; k += 1024) {
        float wr[HEAD_CP], g[HEAD_CP];
#pragma unroll
        for (int o = 0; o < HEAD_CP; ++o) { wr[o] = o < C ? w[(size_t)o * In + k] : 0.f; g[o] = 0.f; }
        // the batch rows in chunks of RC: the next chunk's x values are in flight while this one is multiplied (a fully
        // unrolled loop over 32 rows made the compiler hoist all 512 LDS reads: 536 spilled registers)
        constexpr int RC = 4;
        float xc[RC], xn[RC];
#pragma unroll
        for (int j = 0; j < RC; ++j) xc[j] = j < B ? x[(size_t)j * In + k] : 0.f;
#pragma unroll 1
        for (int b0 = 0; b0 < 32; b0 += RC) {
#pragma unroll
            for (int j = 0; j < RC; ++j) xn[j] = b0 + RC + j < B ? x[(size_t)(b0 + RC + j) * In + k] : 0.f;
#pragma unroll
            for (int j = 0; j < RC; ++j) {
                const int b = b0 + j;
                float dl[HEAD_CP];
#pragma unroll
                for (int q = 0; q < HEAD_CP / 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&s_dl[b][4 * q]);   // same address in every lane: broadcast
                    dl[4 * q] = t[0]; dl[4 * q + 1] = t[1]; dl[4 * q + 2] = t[2]; dl[4 * q + 3] = t[3];
                }
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < HEAD_CP; ++o) { d += dl[o] * wr[o]; g[o] += dl[o] * xc[j]; }
                if (dx != nullptr && b < B) dx[(size_t)b * In + k] = (a.relu_mask && !(xc[j] > 0.f)) ? 0.f : d;
            }
#pragma unroll
            for (int j = 0; j < RC; ++j) xc[j] = xn[j];
        }
        if (a.opt.kind) {
#pragma unroll
            for (int o = 0; o < HEAD_CP; ++o)
                if (o < C) a.opt.weight(sc, c, (long)o * In + k, g[o]);
        } else {
            float* dw = a.dw + c * a.dw_cs;
#pragma unroll
            for (int o = 0; o < HEAD_CP; ++o)
                if (o < C) dw[(size_t)o * In + k] = g[o];
        }
    }
    if (threadIdx.x < C) {
        float s = 0.f;
        for (int b = 0; b < 32; ++b) s += s_dl[b][threadIdx.x];
        if (a.opt.kind) a.opt.bias(sc, c, threadIdx.x, s);
        else a.db[c * a.db_cs + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------
// The same head as TWO kernels over (client, column slice) grids. One workgroup per client leaves most of the chip
// idle (50 workgroups of a client group, 10 of a small shard) and its three phases are chains of memory latencies:
// 107 us per group step of the headline round, on the critical path (profiles/r3_groups_pmc_after_summary.csv).
// A slice cannot simply take its columns of all three phases in one kernel - it would update its columns of W in
// place while another slice of the client still reads them for the logits - so the kernel boundary is the barrier:
//   head_logits_kernel : slice s of client c reduces ITS columns of x W^T -> raw partial tile (accumulator layout)
//   head_tail_kernel   : every slice adds the S partial tiles in the same fixed order (bitwise the same logits in all
//                        of them, reproducible), repeats the tiny softmax / gradient of the 32 x C logits, and then owns
//                        its columns: data gradient, weight gradient, optimizer step. Slice 0 also keeps the counters.
// S depends on the layer width only (never on the number of clients: shards of different sizes, e.g. the client
// groups of one round, must produce the same bits).
// ---------------------------------------------------------------------------------------
constexpr int HEAD_SLICE_COLS = 256;

__global__ __launch_bounds__(256) void head_logits_kernel(HeadArgs a, float* part) {
    __shared__ float red[4][16][64];
    constexpr int U = 4;
    const int c = blockIdx.y, S = gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, kq = (lane >> 5) * 8;
    LoadRowMajorK xl{a.x + c * a.x_cs, a.B, a.In, a.In, (bool)(a.vec & 1)};
    LoadRowMajorK wl{a.w + c * a.w_cs, a.C, a.In, a.In, (bool)(a.vec & 2)};
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nsteps_all = (a.In + 15) / 16;
    const int per = (nsteps_all + S - 1) / S;
    const int s_begin = blockIdx.x * per;
    const int s_end = min(nsteps_all, s_begin + per);
    for (int s0 = s_begin + wave; s0 < s_end; s0 += 4 * U) {
        float xv[U][8], wv[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (s0 + 4 * u < s_end) ? (s0 + 4 * u) * 16 + kq : a.In;     // past the slice: zeros
            xl.load8(r, k, xv[u]);
            wl.load8(r, k, wv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u][e], wv[u][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[e];
    __syncthreads();
    float* dst = part + ((size_t)c * S + blockIdx.x) * 1024;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int e = i >> 6, ln = i & 63;
        dst[i] = (red[0][e][ln] + red[1][e][ln]) + (red[2][e][ln] + red[3][e][ln]);
    }
}

template <int HEAD_CP, int RC>
__global__ __launch_bounds__(256) void head_tail_kernel(HeadArgs a, const float* part) {
    __shared__ __attribute__((aligned(16))) float s_lg[32][HEAD_CMAX];
    __shared__ __attribute__((aligned(16))) float s_dl[32][HEAD_CMAX];
    __shared__ float s_loss[4];
    __shared__ int s_corr[4];
    __shared__ int s_y[32];

    const int c = blockIdx.y, S = gridDim.x;
    const bool keeper = blockIdx.x == 0;               // the slice that owns the counters, the logits output and the bias
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* x = a.x + c * a.x_cs;
    float* w = a.w + c * a.w_cs;
    const float* bias = a.b + c * a.b_cs;
    const int B = a.B, In = a.In, C = a.C;

    // ---- logits = sum of the S partial tiles, slice order; the labels of the batch rows (two dependent loads) fly with them
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], B) : B;
    if (threadIdx.x < 32 && (int)threadIdx.x < nv) {
        const int sample = a.sidx ? a.sidx[c * a.s_cs + threadIdx.x] : (int)threadIdx.x;
        s_y[threadIdx.x] = a.labels[c * a.y_cs + sample];
    }
    {
        const float* src = part + (size_t)c * S * 1024;
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const int e = i >> 6, ln = i & 63, n = ln & 31;
            if (n >= HEAD_CP) continue;
            float pv[16];                              // S <= 16 tiles, all requested before the first add
#pragma unroll
            for (int s = 0; s < 16; ++s) pv[s] = s < S ? src[(size_t)s * 1024 + i] : 0.f;
            float v = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) v += pv[s];   // (+ 0.f past S: the same bits)
            s_lg[mfma_row(e, ln)][n] = n < C ? v + bias[n] : 0.f;
        }
    }
    __syncthreads();

    // ---- softmax cross-entropy; wave w takes rows w, w + 4, ...; a lane owns one class
    {
        const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
        float lsum = 0.f; int lcorrect = 0;           // wave-uniform
        for (int b = wave; b < 32; b += 4) {
            if (b >= nv) {
                if (lane < HEAD_CP) s_dl[b][lane] = 0.f;
                continue;
            }
            const int y = s_y[b];
            const float v = lane < C ? s_lg[b][lane] : -INFINITY;
            if (keeper && a.logits && lane < C) a.logits[c * a.lo_cs + (size_t)b * C + lane] = v;
            float mx = v; int am = lane < C ? lane : 0x7fffffff;       // max and FIRST arg-max
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float om = __shfl_xor(mx, off, 64);
                const int oa = __shfl_xor(am, off, 64);
                if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
            }
            const float ex = lane < C ? expf(v - mx) : 0.f;
            const float se = wave_reduce_sum(ex);
            const float zy = __shfl(v, y, 64);
            lsum += mx + logf(se) - zy;
            lcorrect += (am == y);
            if (lane < HEAD_CP) s_dl[b][lane] = lane < C ? (ex / se - (lane == y ? 1.f : 0.f)) * inv : 0.f;
            if (keeper && a.confusion && lane == 0) atomicAdd(a.confusion + ((size_t)c * C + y) * C + am, 1);
        }
        if (lane == 0) { s_loss[wave] = lsum; s_corr[wave] = lcorrect; }
    }
    __syncthreads();
    if (keeper && threadIdx.x == 0) {
        a.loss_sum[c] += (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        a.correct[c] += s_corr[0] + s_corr[1] + s_corr[2] + s_corr[3];
        a.count[c] += nv;
    }

    // ---- my columns: a thread owns column k of W - data gradient, weight gradient, optimizer step
    const OptEpi::Scal sc = a.opt.kind ? a.opt.scal() : OptEpi::Scal{0.f, 0.f};
    float* dx = a.dx ? a.dx + c * a.dx_cs : nullptr;
    const int per = ((In + 15) / 16 + S - 1) / S * 16;             // the columns head_logits_kernel gave this slice
    const int k_end = min(In, (int)(blockIdx.x + 1) * per);
    for (int k = blockIdx.x * per + threadIdx.x; k < k_end; k += 256) {
        // every global operand of the column is requested before the first use: one memory latency instead of a chain
        float wr[HEAD_CP], g[HEAD_CP], mr[HEAD_CP], vr[HEAD_CP];
        const long wo = (long)c * a.opt.cs + k;
#pragma unroll
        for (int o = 0; o < HEAD_CP; ++o) {
            wr[o] = o < C ? w[(size_t)o * In + k] : 0.f;
            g[o] = 0.f; mr[o] = 0.f; vr[o] = 0.f;
            if (a.opt.kind && o < C) {
                mr[o] = a.opt.mw[wo + (long)o * In];
                if (a.opt.kind == 1) vr[o] = a.opt.vw[wo + (long)o * In];
            }
        }
        // the batch rows in two halves of RC: the second half is requested before the first is used (and the loop is
        // kept rolled: fully unrolled the compiler hoists all 32 x C broadcast reads of the gradient and spills)
        float xc[RC], xn[RC];
#pragma unroll
        for (int j = 0; j < RC; ++j) xc[j] = j < B ? x[(size_t)j * In + k] : 0.f;
#pragma unroll 1
        for (int b0 = 0; b0 < 32; b0 += RC) {
#pragma unroll
            for (int j = 0; j < RC; ++j) xn[j] = b0 + RC + j < B ? x[(size_t)(b0 + RC + j) * In + k] : 0.f;
#pragma unroll
            for (int j = 0; j < RC; ++j) {
                const int b = b0 + j;
                float dl[HEAD_CP];
#pragma unroll
                for (int q = 0; q < HEAD_CP / 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&s_dl[b][4 * q]);   // same address in every lane: broadcast
                    dl[4 * q] = t[0]; dl[4 * q + 1] = t[1]; dl[4 * q + 2] = t[2]; dl[4 * q + 3] = t[3];
                }
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < HEAD_CP; ++o) { d += dl[o] * wr[o]; g[o] += dl[o] * xc[j]; }
                if (dx != nullptr && b < B) dx[(size_t)b * In + k] = (a.relu_mask && !(xc[j] > 0.f)) ? 0.f : d;
            }
#pragma unroll
            for (int j = 0; j < RC; ++j) xc[j] = xn[j];
        }
        if (a.opt.kind) {
#pragma unroll
            for (int o = 0; o < HEAD_CP; ++o)
                if (o < C) {
                    const long e = wo + (long)o * In;
                    a.opt.apply_loaded(sc, a.opt.pw + e, a.opt.mw + e, a.opt.vw ? a.opt.vw + e : nullptr, wr[o], mr[o], vr[o], g[o]);
                }
        } else {
            float* dw = a.dw + c * a.dw_cs;
#pragma unroll
            for (int o = 0; o < HEAD_CP; ++o)
                if (o < C) dw[(size_t)o * In + k] = g[o];
        }
    }
    if (keeper && threadIdx.x < C) {
        float sb = 0.f;
        for (int b = 0; b < 32; ++b) sb += s_dl[b][threadIdx.x];
        if (a.opt.kind) a.opt.bias(sc, c, threadIdx.x, sb);
        else a.db[c * a.db_cs + threadIdx.x] = sb;
    }
}

// column slices of the two-kernel form (1: the one-kernel form); a function of the layer width ONLY
int head_slices(int In) {
    static const bool off = [] { const char* e = getenv("P2PFL_HEAD_SPLIT"); return e && atoi(e) == 0; }();
    if (off || In < 4 * HEAD_SLICE_COLS) return 1;
    const int s = In / HEAD_SLICE_COLS;
    return s > 16 ? 16 : s;
}

}  // namespace

// 1 when p2pfl_linear_head covers this shape (the caller otherwise runs forward / loss / gradients as separate kernels)
P2PFL_API int p2pfl_linear_head_accepts(int B, int In, int C) {
    return B >= 1 && B <= 32 && C >= 2 && C <= HEAD_CMAX && In >= 16;
}

// floats of scratch p2pfl_linear_head wants for this problem (0: the one-kernel form, no scratch)
P2PFL_API long p2pfl_linear_head_scratch(int K, int B, int In, int C) {
    if (K <= 0 || !p2pfl_linear_head_accepts(B, In, C)) return 0;
    const int S = head_slices(In);
    return S > 1 ? (long)K * S * 1024 : 0;
}

P2PFL_API int p2pfl_linear_head(const float* x, long x_cs, float* w, long w_cs, float* b, long b_cs, float* dx,
                                long dx_cs, float* dw, long dw_cs, float* db, long db_cs, float* logits, long lo_cs,
                                const int* labels, long y_cs, const int* sidx, long s_cs, float* loss_sum,
                                int* correct, int* count, int* confusion, const int* n_valid, long nv_cs, int K, int B,
                                int In, int C, int relu_mask, const P2pflOpt* opt, float* scratch, long scratch_floats,
                                hipStream_t stream) {
    if (K <= 0 || !p2pfl_linear_head_accepts(B, In, C)) return -1;
    if (!opt_epi_valid(opt)) return -1;
    const bool fold = opt != nullptr && opt->kind != 0;
    if (!fold && (dw == nullptr || db == nullptr)) return -1;
    if (fold && (opt->pw != w || opt->pb != b)) return -2;                        // the update is in place
    if (fold && K > 1 && (opt->cs != w_cs || b_cs != w_cs)) return -2;
    const int xv = !((In & 3) || (x_cs & 3) || ((uintptr_t)x & 15));
    const int wv = !((In & 3) || (w_cs & 3) || ((uintptr_t)w & 15));
    HeadArgs a{x, x_cs, w, w_cs, b, b_cs, dx, dx_cs, dw, dw_cs, db, db_cs, logits, lo_cs, labels, y_cs, sidx, s_cs,
               loss_sum, correct, count, confusion, n_valid, nv_cs, B, In, C, relu_mask, xv | (wv << 1),
               make_opt_epi(opt)};
    const int S = head_slices(In);
    if (S > 1 && scratch != nullptr && scratch_floats >= (long)K * S * 1024 && K <= 65535) {
        const dim3 grid(S, K);
        hipLaunchKernelGGL(head_logits_kernel, grid, dim3(256), 0, stream, a, scratch);
        static const int rc = [] { const char* e = getenv("P2PFL_HEAD_RC"); return e ? atoi(e) : 4; }();
#define P2PFL_TAIL(CP_)                                                                                                     \
    do {                                                                                                                    \
        if (rc == 8) hipLaunchKernelGGL((head_tail_kernel<CP_, 8>), grid, dim3(256), 0, stream, a, (const float*)scratch);   \
        else if (rc == 16) hipLaunchKernelGGL((head_tail_kernel<CP_, 16>), grid, dim3(256), 0, stream, a, (const float*)scratch); \
        else hipLaunchKernelGGL((head_tail_kernel<CP_, 4>), grid, dim3(256), 0, stream, a, (const float*)scratch);           \
    } while (0)
        if (C <= 4) P2PFL_TAIL(4);
        else if (C <= 8) P2PFL_TAIL(8);
        else if (C <= 12) P2PFL_TAIL(12);
        else P2PFL_TAIL(16);
#undef P2PFL_TAIL
        P2PFL_RETURN_LAST_ERROR();
    }
    // k-steps in flight per wave: 4 while a wave has >= 4 of them (In >= 1024), else 1
#define P2PFL_HEAD(CP_)                                                                                   \
    do {                                                                                                  \
        if (In >= 1024) hipLaunchKernelGGL((linear_head_kernel<4, CP_>), dim3(K), dim3(1024), 0, stream, a); \
        else hipLaunchKernelGGL((linear_head_kernel<1, CP_>), dim3(K), dim3(1024), 0, stream, a);         \
    } while (0)
    if (C <= 4) P2PFL_HEAD(4);
    else if (C <= 8) P2PFL_HEAD(8);
    else if (C <= 12) P2PFL_HEAD(12);
    else P2PFL_HEAD(16);
#undef P2PFL_HEAD
    P2PFL_RETURN_LAST_ERROR();
}
