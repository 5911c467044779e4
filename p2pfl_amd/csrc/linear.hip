// Client-batched fully-connected layers on the gfx950 matrix cores.
//
// One launch covers every co-resident federation client: blockIdx.z is the client, each
// client has its *own* weights (rows of the [clients, P] parameter slab) and its own
// activations. Replaces the per-node nn.Linear calls of the reference models
// (reference: fedstellar/learning/pytorch/mnist/models/mlp.py:156-160,
//  mnist/models/cnn.py:143-144,157-158, femnist/models/cnn.py:140,151).
#include "gemm_core.h"

namespace {

struct LinearArgs {
    const float* x;  long x_cs;   // [K][B][In]   input activations
    const float* w;  long w_cs;   // [K][Out][In] weights (slab view: base+offset, stride = slab row)
    const float* b;  long b_cs;   // [K][Out]     bias
    float* y;        long y_cs;   // [K][B][Out]
    int B, In, Out, relu, vec;
};

struct EpiBiasAct {
    float* y; const float* bias; int M, N, relu;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n >= N) return;
        const float bv = bias[n];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < M) {
                float v = acc[e] + bv;
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)m * N + n] = v;
            }
        }
    }
};

// y = act(x W^T + b);  M = B, N = Out, K = In
__global__ __launch_bounds__(256) void linear_fwd_kernel(LinearArgs a) {
    __shared__ GemmSmem<32, 128> sm;
    const int c = blockIdx.z;
    const float* x = a.x + c * a.x_cs;
    const float* w = a.w + c * a.w_cs;
    LoadRowMajorK al{x, a.B, a.In, a.In, (bool)(a.vec & 1)};
    LoadRowMajorK bl{w, a.Out, a.In, a.In, (bool)(a.vec & 2)};
    EpiBiasAct epi{a.y + c * a.y_cs, a.b + c * a.b_cs, a.B, a.Out, a.relu};
    gemm_block_tile<32, 128, 1, 4>(sm, al, bl, epi, blockIdx.y * 32, blockIdx.x * 128, 0, a.In);
}

struct DgradArgs {
    const float* dy; long dy_cs;   // [K][B][Out]
    const float* w;  long w_cs;    // [K][Out][In]
    const float* act; long act_cs; // [K][B][In] post-ReLU activation of the producer layer (or null)
    float* dx;       long dx_cs;   // [K][B][In]
    int B, In, Out, vec;
};

struct EpiMaskStore {
    float* dx; const float* act; int M, N;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n >= N) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < M) {
                float v = acc[e];
                if (act != nullptr && !(act[(size_t)m * N + n] > 0.f)) v = 0.f;
                dx[(size_t)m * N + n] = v;
            }
        }
    }
};

// dx = (dy W) * [act > 0];  M = B, N = In, K = Out
__global__ __launch_bounds__(256) void linear_dgrad_kernel(DgradArgs a) {
    __shared__ GemmSmem<32, 128> sm;
    const int c = blockIdx.z;
    LoadRowMajorK al{a.dy + c * a.dy_cs, a.B, a.Out, a.Out, (bool)a.vec};
    LoadColMajorK bl{a.w + c * a.w_cs, a.In, a.Out, a.In};
    EpiMaskStore epi{a.dx + c * a.dx_cs, a.act ? a.act + c * a.act_cs : nullptr, a.B, a.In};
    gemm_block_tile<32, 128, 1, 4>(sm, al, bl, epi, blockIdx.y * 32, blockIdx.x * 128, 0, a.Out);
}

struct WgradArgs {
    const float* dy; long dy_cs;  // [K][B][Out]
    const float* x;  long x_cs;   // [K][B][In]
    float* dw;       long dw_cs;  // [K][Out][In]  (grad slab view)
    float* db;       long db_cs;  // [K][Out]
    int B, In, Out;
};

// x^T with one extra all-ones row at index In: the bias gradient drops out of the same
// MFMA pass as column In of the product.
struct LoadXTOnes {
    static constexpr bool ROW_CONTIG = true;
    const float* x; int In, B;
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = 0.f;
            if (k + j < B) {
                if (row < In) t = x[(size_t)(k + j) * In + row];
                else if (row == In) t = 1.f;
            }
            v[j] = t;
        }
    }
};

struct EpiWgradStore {
    float* dw; float* db; int M, N;  // M = Out, N = In (column N = bias)
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

// dW = dy^T x, db = dy^T 1;  M = Out, N = In + 1, K = B
__global__ __launch_bounds__(256) void linear_wgrad_kernel(WgradArgs a) {
    __shared__ GemmSmem<128, 128> sm;
    const int c = blockIdx.z;
    LoadColMajorK al{a.dy + c * a.dy_cs, a.Out, a.B, a.Out};
    LoadXTOnes bl{a.x + c * a.x_cs, a.In, a.B};
    EpiWgradStore epi{a.dw + c * a.dw_cs, a.db + c * a.db_cs, a.Out, a.In};
    gemm_block_tile<128, 128, 2, 2>(sm, al, bl, epi, blockIdx.y * 128, blockIdx.x * 128, 0, a.B);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

P2PFL_API int p2pfl_linear_fwd(const float* x, long x_cs, const float* w, long w_cs, const float* b,
                               long b_cs, float* y, long y_cs, int K, int B, int In, int Out,
                               int relu, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    // float4 loads need In % 4 == 0 and 16-B aligned bases/strides; otherwise element loads
    const int xv = !((In & 3) || (x_cs & 3) || ((uintptr_t)x & 15));
    const int wv = !((In & 3) || (w_cs & 3) || ((uintptr_t)w & 15));
    LinearArgs a{x, x_cs, w, w_cs, b, b_cs, y, y_cs, B, In, Out, relu, xv | (wv << 1)};
    dim3 grid(cdiv(Out, 128), cdiv(B, 32), K);
    hipLaunchKernelGGL(linear_fwd_kernel, grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_linear_dgrad(const float* dy, long dy_cs, const float* w, long w_cs,
                                 const float* act, long act_cs, float* dx, long dx_cs, int K, int B,
                                 int In, int Out, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    // dy rows of e.g. 10 classes are not 16-B aligned: element loads then
    const int vec = !((Out & 3) || (dy_cs & 3) || ((uintptr_t)dy & 15));
    DgradArgs a{dy, dy_cs, w, w_cs, act, act_cs, dx, dx_cs, B, In, Out, vec};
    dim3 grid(cdiv(In, 128), cdiv(B, 32), K);
    hipLaunchKernelGGL(linear_dgrad_kernel, grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_linear_wgrad(const float* dy, long dy_cs, const float* x, long x_cs, float* dw,
                                 long dw_cs, float* db, long db_cs, int K, int B, int In, int Out,
                                 hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    WgradArgs a{dy, dy_cs, x, x_cs, dw, dw_cs, db, db_cs, B, In, Out};
    dim3 grid(cdiv(In + 1, 128), cdiv(Out, 128), K);
    hipLaunchKernelGGL(linear_wgrad_kernel, grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}
