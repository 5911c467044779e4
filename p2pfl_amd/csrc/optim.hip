// Fused multi-client optimizers over the [clients, P] slabs (param / grad / moment rows).
// One launch updates every parameter of every co-resident client: a pure HBM stream
// (float4 per lane, grid-stride), replacing the per-tensor torch.optim loops the reference
// runs once per node (reference: fedstellar/learning/pytorch/mnist/models/cnn.py:161-164
// Adam(lr=1e-3); femnist/models/cnn.py:153-157 SGD(lr=1e-3, momentum=0.9)).
#include "common.h"

namespace {

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    long cs;            // row stride (floats) shared by the parameter / moment slabs
    long g_cs;          // row stride of the gradient slab (the engine's leaves out the layers with fused updates)
    long n4;            // float4 elements per row
    long n4a, skip4;    // two column ranges in one launch: element i >= n4a lives skip4 float4 further on (one range: n4a = n4)
    float lr, beta1, beta2, eps, weight_decay;
    float inv_bc1;      // 1 / (1 - beta1^t)
    float inv_sqrt_bc2; // 1 / sqrt(1 - beta2^t)
    const float* bc;    // optional device copy of {inv_bc1, inv_sqrt_bc2} (graph replay with varying t)
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const long row = (long)blockIdx.y * a.cs;
    float4* p = reinterpret_cast<float4*>(a.p + row);
    const float4* g = reinterpret_cast<const float4*>(a.g + (long)blockIdx.y * a.g_cs);
    float4* m = reinterpret_cast<float4*>(a.m + row);
    float4* v = reinterpret_cast<float4*>(a.v + row);
    const float inv_bc1 = a.bc ? a.bc[0] : a.inv_bc1;
    const float inv_sqrt_bc2 = a.bc ? a.bc[1] : a.inv_sqrt_bc2;
    const float step = a.lr * inv_bc1;
    const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < a.n4; i0 += (long)gridDim.x * 256) {
        const long i = i0 < a.n4a ? i0 : i0 + a.skip4;
        float4 pv = p[i], gv = g[i], mv = m[i], vv = v[i];
        float* pp = &pv.x; float* gg = &gv.x; float* mm = &mv.x; float* vq = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] + a.weight_decay * pp[j];
            mm[j] = a.beta1 * mm[j] + omb1 * gr;
            vq[j] = a.beta2 * vq[j] + omb2 * gr * gr;
            const float denom = sqrtf(vq[j]) * inv_sqrt_bc2 + a.eps;
            pp[j] -= step * (mm[j] / denom);
        }
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

struct SgdArgs {
    float* p; const float* g; float* buf;
    long cs, g_cs, n4;
    float lr, momentum, weight_decay;
};

__global__ __launch_bounds__(256) void sgdm_kernel(SgdArgs a) {
    const long row = (long)blockIdx.y * a.cs;
    float4* p = reinterpret_cast<float4*>(a.p + row);
    const float4* g = reinterpret_cast<const float4*>(a.g + (long)blockIdx.y * a.g_cs);
    float4* b = reinterpret_cast<float4*>(a.buf + row);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (long)gridDim.x * 256) {
        float4 pv = p[i], gv = g[i], bv = b[i];
        float* pp = &pv.x; float* gg = &gv.x; float* bb = &bv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] + a.weight_decay * pp[j];
            bb[j] = a.momentum * bb[j] + gr;   // zero-initialised buffer == torch's first-step copy
            pp[j] -= a.lr * bb[j];
        }
        p[i] = pv; b[i] = bv;
    }
}

inline int grid_x(long n4, int K) {
    long want = (n4 + 255) / 256;
    long cap = (256L * 8 + K - 1) / K;   // ~2048 blocks over the whole launch
    if (cap < 1) cap = 1;
    return (int)(want < cap ? want : cap);
}

}  // namespace

P2PFL_API int p2pfl_adam_step(float* p, const float* g, float* m, float* v, long cs, long g_cs, long n, int K,
                              float lr, float beta1, float beta2, float eps, float weight_decay,
                              int t, const float* bc, hipStream_t stream) {
    if (K <= 0 || n <= 0 || t <= 0) return -1;
    if ((n & 3) || (cs & 3) || (g_cs & 3) || ((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) ||
        ((uintptr_t)v & 15))
        return -2;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    AdamArgs a{p, g, m, v, cs, g_cs, n / 4, n / 4, 0, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
               (float)(1.0 / sqrt(bc2)), bc};
    hipLaunchKernelGGL(adam_kernel, dim3(grid_x(n / 4, K), K), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

// Two column ranges of the slabs in ONE launch: [0, n0) and [n0 + gap, n0 + gap + n1) relative to the pointers (the
// parameters in front of and behind a layer whose update is fused into its weight-gradient kernel). One launch instead
// of two matters where the launch shares the GPU with a kernel that leaves no free wave slots: every launch then waits
// for that kernel's workgroups to drain before its own get on (19 -> 270 us beside the fused update).
P2PFL_API int p2pfl_adam_step2(float* p, const float* g, float* m, float* v, long cs, long n0, long gap, long n1, int K,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               int t, const float* bc, hipStream_t stream) {
    if (K <= 0 || n0 <= 0 || n1 <= 0 || gap < 0 || t <= 0) return -1;
    if ((n0 & 3) || (n1 & 3) || (gap & 3) || (cs & 3) || ((uintptr_t)p & 15) || ((uintptr_t)g & 15) ||
        ((uintptr_t)m & 15) || ((uintptr_t)v & 15))
        return -2;
    if (n0 + gap + n1 > cs) return -3;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    AdamArgs a{p, g, m, v, cs, cs, (n0 + n1) / 4, n0 / 4, gap / 4, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
               (float)(1.0 / sqrt(bc2)), bc};
    hipLaunchKernelGGL(adam_kernel, dim3(grid_x((n0 + n1) / 4, K), K), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_sgdm_step(float* p, const float* g, float* buf, long cs, long g_cs, long n, int K, float lr,
                              float momentum, float weight_decay, hipStream_t stream) {
    if (K <= 0 || n <= 0) return -1;
    if ((n & 3) || (cs & 3) || (g_cs & 3) || ((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)buf & 15))
        return -2;
    SgdArgs a{p, g, buf, cs, g_cs, n / 4, lr, momentum, weight_decay};
    hipLaunchKernelGGL(sgdm_kernel, dim3(grid_x(n / 4, K), K), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}
