// Fused (log-)softmax + cross-entropy forward/backward, accuracy counters and confusion matrix
// for every client of the shard in one launch (one 64-lane wave per client-batch of <= 64 rows
// would under-fill the chip, so a block handles one client and each *lane* owns one sample row:
// C <= 64 classes are walked serially in registers; the work is a few kFLOP and purely
// launch-bound, which is why it is fused into one kernel at all).
//
// Semantics (reference: fedstellar/learning/pytorch/mnist/models/mlp.py:142,161,169-180 and
// mnist/models/cnn.py:166-177): loss = mean_b CE(logits_b, y_b). The MLP applies log_softmax
// before CrossEntropyLoss; log_softmax is idempotent, so one kernel serves both models.
#include "common.h"

namespace {

struct CeArgs {
    const float* logits; long l_cs;     // [K][B][C]
    const int* labels;   long y_cs;     // [K][n_samples]
    const int* sidx;     long s_cs;     // [K][B] sample gather (or null: identity)
    float* dlogits;      long d_cs;     // [K][B][C]  (null on the evaluation path)
    float* loss_sum;                    // [K]  += sum_b loss_b
    int* correct;                       // [K]  += #(argmax == y)
    int* count;                         // [K]  += n_valid
    int* confusion;                     // [K][C][C] (row = target, col = prediction) or null
    const int* n_valid;  long nv_cs;    // [K] rows of this batch that are real samples (or null: B)
    int B, C;
};

__global__ __launch_bounds__(64) void softmax_ce_kernel(CeArgs a) {
    const int c = blockIdx.x;
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], a.B) : a.B;
    const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
    float lsum = 0.f; int lcorrect = 0;
    for (int b = threadIdx.x; b < a.B; b += 64) {
        const float* z = a.logits + c * a.l_cs + (size_t)b * a.C;
        float* dz = a.dlogits ? a.dlogits + c * a.d_cs + (size_t)b * a.C : nullptr;
        if (b >= nv) {
            if (dz) for (int j = 0; j < a.C; ++j) dz[j] = 0.f;
            continue;
        }
        const int sample = a.sidx ? a.sidx[c * a.s_cs + b] : b;
        const int y = a.labels[c * a.y_cs + sample];
        float mx = z[0]; int am = 0;
        for (int j = 1; j < a.C; ++j) { const float v = z[j]; if (v > mx) { mx = v; am = j; } }
        float se = 0.f;
        for (int j = 0; j < a.C; ++j) se += expf(z[j] - mx);
        const float lse = mx + logf(se);
        lsum += lse - z[y];
        lcorrect += (am == y);
        if (dz) {
            const float rs = 1.f / se;
            for (int j = 0; j < a.C; ++j) {
                const float p = expf(z[j] - mx) * rs;
                dz[j] = (p - (j == y ? 1.f : 0.f)) * inv;
            }
        }
        if (a.confusion) atomicAdd(a.confusion + ((size_t)c * a.C + y) * a.C + am, 1);
    }
    lsum = wave_reduce_sum(lsum);
    float fc = wave_reduce_sum((float)lcorrect);
    if (threadIdx.x == 0) {
        a.loss_sum[c] += lsum;
        a.correct[c] += (int)(fc + 0.5f);
        a.count[c] += nv;
    }
}

// One logit per sample (C == 1): sigmoid + binary cross-entropy, the loss of the reference's WADI model
// (fedstellar/learning/pytorch/wadi/models/mlp.py: torch.sigmoid in forward, BCELoss) - the two-class softmax over the
// logits (0, z). loss = softplus(z) - y z computed in its stable form, d loss / dz = (sigmoid(z) - y) / n_valid, the
// prediction is z > 0 (p > 0.5); the confusion matrix is 2 x 2.
__global__ __launch_bounds__(64) void sigmoid_bce_kernel(CeArgs a) {
    const int c = blockIdx.x;
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], a.B) : a.B;
    const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
    float lsum = 0.f; int lcorrect = 0;
    for (int b = threadIdx.x; b < a.B; b += 64) {
        float* dz = a.dlogits ? a.dlogits + c * a.d_cs + b : nullptr;
        if (b >= nv) {
            if (dz) *dz = 0.f;
            continue;
        }
        const int sample = a.sidx ? a.sidx[c * a.s_cs + b] : b;
        const int y = a.labels[c * a.y_cs + sample] != 0;
        const float z = a.logits[c * a.l_cs + b];
        const float e = expf(-fabsf(z));                       // in (0, 1]: no overflow either way
        // (BCELoss clamps its logarithms at -100, i.e. the loss of one sample at 100)
        lsum += fminf(fmaxf(z, 0.f) - (y ? z : 0.f) + log1pf(e), 100.f);
        const float p = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        const int pred = p > 0.5f;
        lcorrect += (pred == y);
        if (dz) *dz = (p - (float)y) * inv;
        if (a.confusion) atomicAdd(a.confusion + ((size_t)c * 2 + y) * 2 + pred, 1);
    }
    lsum = wave_reduce_sum(lsum);
    float fc = wave_reduce_sum((float)lcorrect);
    if (threadIdx.x == 0) {
        a.loss_sum[c] += lsum;
        a.correct[c] += (int)(fc + 0.5f);
        a.count[c] += nv;
    }
}

// Reconstruction loss of the auto-encoders (reference syscall/models/autoencoder.py: MSELoss between the decoder
// output and the INPUT batch): loss = mean over the n_valid x D elements of (out - x)^2, d loss / d out =
// 2 (out - x) / (n_valid D); x rows are gathered from the client's data with the batch's sample indices. One block
// per client; the block's sum is formed in a fixed order (bitwise reproducible).
struct MseArgs {
    const float* out; long o_cs;        // [K][B][D]
    const float* x;   long x_cs;        // [K][n][D]
    const int* sidx;  long s_cs;        // [K][B] (or null: rows 0 .. B-1)
    float* dout;      long d_cs;        // [K][B][D] (null on the evaluation path)
    float* loss_sum; int* count;
    const int* n_valid; long nv_cs;
    int B, D;
};

__global__ __launch_bounds__(256) void mse_input_kernel(MseArgs a) {
    __shared__ float s_part[4];
    const int c = blockIdx.x;
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], a.B) : a.B;
    const float inv = nv > 0 ? 1.f / ((float)nv * (float)a.D) : 0.f;
    float l = 0.f;
    for (int i = threadIdx.x; i < a.B * a.D; i += 256) {
        const int b = i / a.D, j = i - b * a.D;
        float* dz = a.dout ? a.dout + c * a.d_cs + i : nullptr;
        if (b >= nv) {
            if (dz) *dz = 0.f;
            continue;
        }
        const int row = a.sidx ? a.sidx[c * a.s_cs + b] : b;
        const float d = a.out[c * a.o_cs + i] - a.x[c * a.x_cs + (size_t)row * a.D + j];
        l += d * d;
        if (dz) *dz = 2.f * d * inv;
    }
    l = wave_reduce_sum(l);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sum = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
        a.loss_sum[c] += sum * inv * (float)nv;        // the engines accumulate (mean loss of the batch) x (its samples)
        a.count[c] += nv;
    }
}

// Many classes (FEMNIST: 62): a lane per row means three passes of stride-C loads (29 us for 50 clients). Here a
// block of 16 waves serves a client, a wave takes rows w, w + 16, ..., and a lane owns ONE class (C <= 64): a row is
// one coalesced load, the reductions are wave shuffles, the gradient one coalesced store. (For the 10-class models
// the lane-per-row kernel above stays: 9 us against 22 us for this layout.)
constexpr int CE_WAVES = 16;

__global__ __launch_bounds__(64 * CE_WAVES) void softmax_ce_wide_kernel(CeArgs a) {
    __shared__ float s_loss[CE_WAVES];
    __shared__ int s_correct[CE_WAVES];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = a.n_valid ? min(a.n_valid[c * a.nv_cs], a.B) : a.B;
    const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
    float lsum = 0.f; int lcorrect = 0;                       // wave-uniform
    for (int b = wave; b < a.B; b += CE_WAVES) {
        float* dz = a.dlogits ? a.dlogits + c * a.d_cs + (size_t)b * a.C : nullptr;
        if (b >= nv) {
            if (dz && lane < a.C) dz[lane] = 0.f;
            continue;
        }
        const int sample = a.sidx ? a.sidx[c * a.s_cs + b] : b;
        const int y = a.labels[c * a.y_cs + sample];
        const float v = lane < a.C ? a.logits[c * a.l_cs + (size_t)b * a.C + lane] : -INFINITY;
        // max and FIRST arg-max (lowest class index among equal values)
        float mx = v; int am = lane < a.C ? lane : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off, 64);
            const int oa = __shfl_xor(am, off, 64);
            if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
        }
        const float ex = lane < a.C ? expf(v - mx) : 0.f;
        const float se = wave_reduce_sum(ex);
        const float zy = __shfl(v, y, 64);
        lsum += mx + logf(se) - zy;
        lcorrect += (am == y);
        if (dz && lane < a.C) dz[lane] = (ex / se - (lane == y ? 1.f : 0.f)) * inv;
        if (a.confusion && lane == 0) atomicAdd(a.confusion + ((size_t)c * a.C + y) * a.C + am, 1);
    }
    if (lane == 0) { s_loss[wave] = lsum; s_correct[wave] = lcorrect; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f; int k = 0;
#pragma unroll
        for (int w = 0; w < CE_WAVES; ++w) { l += s_loss[w]; k += s_correct[w]; }
        a.loss_sum[c] += l;
        a.correct[c] += k;
        a.count[c] += nv;
    }
}

__global__ void zero_rows_kernel(float* base, long cs, int n, int K) {
    const int c = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        base[c * cs + i] = 0.f;
}

}  // namespace

P2PFL_API int p2pfl_softmax_ce(const float* logits, long l_cs, const int* labels, long y_cs,
                               const int* sidx, long s_cs, float* dlogits, long d_cs,
                               float* loss_sum, int* correct, int* count, int* confusion,
                               const int* n_valid, long nv_cs, int K, int B, int C,
                               hipStream_t stream) {
    if (K <= 0 || B <= 0 || C <= 0) return -1;
    CeArgs a{logits, l_cs, labels, y_cs, sidx, s_cs, dlogits, d_cs, loss_sum, correct, count,
             confusion, n_valid, nv_cs, B, C};
    if (C == 1) hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(K), dim3(64), 0, stream, a);     // confusion: [K][2][2]
    else if (C > 16 && C <= 64) hipLaunchKernelGGL(softmax_ce_wide_kernel, dim3(K), dim3(64 * CE_WAVES), 0, stream, a);
    else hipLaunchKernelGGL(softmax_ce_kernel, dim3(K), dim3(64), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_mse_input(const float* out, long o_cs, const float* x, long x_cs, const int* sidx, long s_cs,
                              float* dout, long d_cs, float* loss_sum, int* count, const int* n_valid, long nv_cs,
                              int K, int B, int D, hipStream_t stream) {
    if (K <= 0 || B <= 0 || D <= 0) return -1;
    MseArgs a{out, o_cs, x, x_cs, sidx, s_cs, dout, d_cs, loss_sum, count, n_valid, nv_cs, B, D};
    hipLaunchKernelGGL(mse_input_kernel, dim3(K), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

// Zero `n` floats at base + c*cs for every client c (gradient regions that are accumulated
// with atomics).
P2PFL_API int p2pfl_zero_rows(float* base, long cs, int n, int K, hipStream_t stream) {
    if (K <= 0 || n <= 0) return -1;
    int bx = (n + 255) / 256; if (bx > 64) bx = 64;
    hipLaunchKernelGGL(zero_rows_kernel, dim3(bx, K), dim3(256), 0, stream, base, cs, n, K);
    P2PFL_RETURN_LAST_ERROR();
}
