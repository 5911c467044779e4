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

}  // namespace

// 1 when p2pfl_linear_head covers this shape (the caller otherwise runs forward / loss / gradients as separate kernels)
P2PFL_API int p2pfl_linear_head_accepts(int B, int In, int C) {
    return B >= 1 && B <= 32 && C >= 2 && C <= HEAD_CMAX && In >= 16;
}

P2PFL_API int p2pfl_linear_head(const float* x, long x_cs, float* w, long w_cs, float* b, long b_cs, float* dx,
                                long dx_cs, float* dw, long dw_cs, float* db, long db_cs, float* logits, long lo_cs,
                                const int* labels, long y_cs, const int* sidx, long s_cs, float* loss_sum,
                                int* correct, int* count, int* confusion, const int* n_valid, long nv_cs, int K, int B,
                                int In, int C, int relu_mask, const P2pflOpt* opt, hipStream_t stream) {
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
