// Optimizer step as the EPILOGUE of the kernel that produces a (small) layer's final gradient.
//
// The big dense layer has had its Adam update fused into its weight-gradient kernel since round 1; the small layers
// (convolution filters, the class head) went through the gradient slab and one element-wise optimizer launch per
// parameter range - 4 extra launches per step on the side stream of the two-stream schedule, each of which waited for
// wave slots beside the HBM-bound update of the other client group (20 us alone, 270 us in the schedule:
// profiles/r2_final_round_timeline.txt). The kernels that hold a layer's FINAL gradient value in a register (the
// fixed-order reduction passes of the convolution weight gradients, the head kernel) now apply the update themselves.
//
// Same arithmetic, operation for operation, as optim.hip (adam_kernel / sgdm_kernel) and as torch.optim.Adam / SGD
// (reference: fedstellar/learning/pytorch/mnist/models/cnn.py:161-164, femnist/models/cnn.py:153-157).
#pragma once
#include "common.h"

// passed BY POINTER through the C ABI (ctypes.Structure in ops/hip.py); null / kind 0 = write the gradient instead
struct P2pflOpt {
    int kind;                   // 0: none, 1: Adam, 2: SGD-momentum
    int t;                      // Adam step (1-based) for the host-side bias corrections when bc == null
    float* pw; float* mw; float* vw;    // weight views of the parameter / first moment (momentum) / second moment slabs
    float* pb; float* mb; float* vb;    // bias views
    long cs;                    // common row (client) stride of all of them, floats
    float lr, beta1, beta2, eps, weight_decay, momentum;
    const float* bc;            // optional device {1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} (graph replay with varying t)
};

struct OptEpi {
    int kind;
    float* pw; float* mw; float* vw;
    float* pb; float* mb; float* vb;
    long cs;
    float lr, beta1, beta2, eps, wd, momentum, inv_bc1, inv_sqrt_bc2;
    const float* bc;

    struct Scal { float step, isb2; };
    __device__ __forceinline__ Scal scal() const {
        const float ib1 = bc ? bc[0] : inv_bc1;
        return Scal{lr * ib1, bc ? bc[1] : inv_sqrt_bc2};
    }
    // one element: parameter / moments at p, m, v (already offset to client and element)
    __device__ __forceinline__ void apply(const Scal& s, float* p, float* m, float* v, float g) const {
        const float pv = *p;
        const float gr = g + wd * pv;
        if (kind == 1) {
            const float mv = beta1 * *m + (1.f - beta1) * gr;
            const float vv = beta2 * *v + (1.f - beta2) * gr * gr;
            *m = mv; *v = vv;
            *p = pv - s.step * (mv / (sqrtf(vv) * s.isb2 + eps));
        } else {
            const float bv = momentum * *m + gr;      // zero-initialised buffer == torch's first-step copy
            *m = bv;
            *p = pv - lr * bv;
        }
    }
    // the same step with the parameter / moment values already in registers (loaded early, beside the operands of the
    // gradient): operation for operation apply()
    __device__ __forceinline__ void apply_loaded(const Scal& s, float* p, float* m, float* v, float pv, float m0, float v0,
                                                 float g) const {
        const float gr = g + wd * pv;
        if (kind == 1) {
            const float mv = beta1 * m0 + (1.f - beta1) * gr;
            const float vv = beta2 * v0 + (1.f - beta2) * gr * gr;
            *m = mv; *v = vv;
            *p = pv - s.step * (mv / (sqrtf(vv) * s.isb2 + eps));
        } else {
            const float bv = momentum * m0 + gr;
            *m = bv;
            *p = pv - lr * bv;
        }
    }
    __device__ __forceinline__ void weight(const Scal& s, int client, long i, float g) const {
        const long o = client * cs + i;
        apply(s, pw + o, mw + o, vw ? vw + o : nullptr, g);
    }
    __device__ __forceinline__ void bias(const Scal& s, int client, long i, float g) const {
        const long o = client * cs + i;
        apply(s, pb + o, mb + o, vb ? vb + o : nullptr, g);
    }
};

#ifndef __HIP_DEVICE_COMPILE__
#include <cmath>
#endif
// host: C ABI struct -> kernel argument (bias corrections in double like the launchers of optim.hip)
inline OptEpi make_opt_epi(const P2pflOpt* o) {
    OptEpi e{};
    if (o == nullptr || o->kind == 0) return e;
    e.kind = o->kind;
    e.pw = o->pw; e.mw = o->mw; e.vw = o->vw; e.pb = o->pb; e.mb = o->mb; e.vb = o->vb; e.cs = o->cs;
    e.lr = o->lr; e.beta1 = o->beta1; e.beta2 = o->beta2; e.eps = o->eps; e.wd = o->weight_decay; e.momentum = o->momentum;
    if (o->kind == 1) {
        const int t = o->t > 0 ? o->t : 1;
        e.inv_bc1 = (float)(1.0 / (1.0 - pow((double)o->beta1, t)));
        e.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)o->beta2, t)));
    }
    e.bc = o->bc;
    return e;
}
inline bool opt_epi_valid(const P2pflOpt* o) {
    if (o == nullptr || o->kind == 0) return true;
    if (o->kind != 1 && o->kind != 2) return false;
    if (!o->pw || !o->mw || !o->pb || !o->mb) return false;
    if (o->kind == 1 && (!o->vw || !o->vb)) return false;
    return true;
}
