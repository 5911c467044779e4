// Client-batched fully-connected layers on the gfx950 matrix cores.
//
// One launch covers every co-resident federation client: blockIdx.z is the client, each
// client has its *own* weights (rows of the [clients, P] parameter slab) and its own
// activations. Replaces the per-node nn.Linear calls of the reference models
// (reference: fedstellar/learning/pytorch/mnist/models/mlp.py:156-160,
//  mnist/models/cnn.py:143-144,157-158, femnist/models/cnn.py:140,151).
#include "gemm_core.h"
#include <type_traits>

namespace {

struct LinearArgs {
    const float* x;  long x_cs;   // [K][B][In]   input activations
    const float* w;  long w_cs;   // [K][Out][In] weights (slab view: base+offset, stride = slab row)
    const float* b;  long b_cs;   // [K][Out]     bias
    float* y;        long y_cs;   // [K][B][Out]
    int B, In, Out, relu, vec;
    const int* sidx; long s_cs;   // optional [K][B] row indices into x (x is then the whole [K][n][In] data set)
    float* part;                  // skinny kernel with gridDim.x > 1: raw partial tiles [K][row tiles][split][NT][16][64]
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
template <int BK, int NBUF, class T>
__global__ __launch_bounds__(256) void linear_fwd_kernel(LinearArgs a) {
    __shared__ GemmSmem<32, 128, NBUF, BK, T> sm;
    const int c = blockIdx.z;
    const float* x = a.x + c * a.x_cs;
    const float* w = a.w + c * a.w_cs;
    LoadRowMajorK al{x, a.B, a.In, a.In, (bool)(a.vec & 1), a.sidx ? a.sidx + c * a.s_cs : nullptr};
    LoadRowMajorK bl{w, a.Out, a.In, a.In, (bool)(a.vec & 2)};
    EpiBiasAct epi{a.y + c * a.y_cs, a.b + c * a.b_cs, a.B, a.Out, a.relu};
    gemm_block_tile<32, 128, 1, 4, LoadRowMajorK, LoadRowMajorK, EpiBiasAct, false, NBUF, BK, T>(
        sm, al, bl, epi, blockIdx.y * 32, blockIdx.x * 128, 0, a.In);
}

// Skinny-output layer (Out <= 64: the 10- / 62-class heads). A 32 x Out x In problem per client has one
// tile, so the LDS pipeline above runs 10..100 blocks that each wait a full HBM latency per k-step (49 us for
// 2048 -> 10). Here the 16 waves of a block split the reduction, fetch their MFMA fragments straight from
// global memory with several k-steps in flight, and add their partial tiles up through LDS.
template <int NT, int U, bool F32>
__global__ __launch_bounds__(1024) void linear_fwd_skinny_kernel(LinearArgs a) {
    __shared__ float red[16][16][64];
    const int c = blockIdx.z, m0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, kq = (lane >> 5) * 8;
    LoadRowMajorK xl{a.x + c * a.x_cs, a.B, a.In, a.In, (bool)(a.vec & 1), a.sidx ? a.sidx + c * a.s_cs : nullptr};
    LoadRowMajorK wl{a.w + c * a.w_cs, a.Out, a.In, a.In, (bool)(a.vec & 2)};
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    // 16-deep k-steps; blockIdx.x owns a contiguous range of them (split reduction over blocks when there are
    // fewer (client, row tile) pairs than CUs), dealt round-robin to the waves
    const int nsteps_all = (a.In + 15) / 16;
    const int per_split = (nsteps_all + gridDim.x - 1) / gridDim.x;
    const int s_begin = blockIdx.x * per_split;
    const int nsteps = min(nsteps_all, s_begin + per_split);
    for (int s0 = s_begin + wave; s0 < nsteps; s0 += 16 * U) {
        float xv[U][8], wv[U][NT][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // past the end of this block's range: zeros (the loaders also return zeros past In)
            const int k = (s0 + 16 * u < nsteps) ? (s0 + 16 * u) * 16 + kq : a.In;
            xl.load8(m0 + r, k, xv[u]);
#pragma unroll
            for (int j = 0; j < NT; ++j) wl.load8(j * 32 + r, k, wv[u][j]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (F32) {      // lane half h holds k = 8h .. 8h+7: MFMA e reduces the pair (e, 8 + e)
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[u][e], wv[u][j][e], acc[j], 0, 0, 0);
            } else {
                const bf16x8 af = pack_bf16x8(xv[u]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pack_bf16x8(wv[u][j]), acc[j], 0, 0, 0);
            }
        }
    }
    float* y = a.y + c * a.y_cs;
    const float* bias = a.b + c * a.b_cs;
    const int e_out = threadIdx.x >> 6;            // after the reduction thread (e, lane) owns one output
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j > 0) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[j][e];
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][e_out][lane];
        if (gridDim.x > 1) {       // raw partial tile; linear_skinny_reduce_kernel finishes the job
            const size_t tile = (((size_t)c * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NT + j;
            a.part[tile * 1024 + threadIdx.x] = v;
            continue;
        }
        const int m = m0 + mfma_row(e_out, lane), n = j * 32 + r;
        if (m < a.B && n < a.Out) {
            v += bias[n];
            if (a.relu) v = fmaxf(v, 0.f);
            y[(size_t)m * a.Out + n] = v;
        }
    }
}

// Sum of the split partial tiles + bias (+ ReLU): one block per (client, row tile), thread = (accumulator, lane).
template <int NT>
__global__ __launch_bounds__(1024) void linear_skinny_reduce_kernel(LinearArgs a, int ksplit) {
    const int c = blockIdx.z, m0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, e_out = threadIdx.x >> 6, r = lane & 31;
    const float* p = a.part + ((size_t)c * gridDim.y + blockIdx.y) * ksplit * NT * 1024 + threadIdx.x;
    float* y = a.y + c * a.y_cs;
    const float* bias = a.b + c * a.b_cs;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float v = 0.f;
        for (int s = 0; s < ksplit; ++s) v += p[((size_t)s * NT + j) * 1024];
        const int m = m0 + mfma_row(e_out, lane), n = j * 32 + r;
        if (m < a.B && n < a.Out) {
            v += bias[n];
            if (a.relu) v = fmaxf(v, 0.f);
            y[(size_t)m * a.Out + n] = v;
        }
    }
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
template <int BK, int NBUF, class T>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(DgradArgs a) {
    __shared__ GemmSmem<32, 128, NBUF, BK, T> sm;
    const int c = blockIdx.z;
    LoadRowMajorK al{a.dy + c * a.dy_cs, a.B, a.Out, a.Out, (bool)a.vec};
    LoadColMajorK bl{a.w + c * a.w_cs, a.In, a.Out, a.In};
    EpiMaskStore epi{a.dx + c * a.dx_cs, a.act ? a.act + c * a.act_cs : nullptr, a.B, a.In};
    gemm_block_tile<32, 128, 1, 4, LoadRowMajorK, LoadColMajorK, EpiMaskStore, false, NBUF, BK, T>(
        sm, al, bl, epi, blockIdx.y * 32, blockIdx.x * 128, 0, a.Out);
}

// ---------------------------------------------------------------------------------------
// Split reduction for shards with too few tiles to fill the chip (10 clients x 16 column tiles = 160 workgroups for
// the MNIST-CNN l1 forward on 256 CUs: 84 us for 257 MB of weights, 3 TB/s). The reduction (In for the forward, Out
// for the data gradient) is cut into S ranges of whole k-steps, one workgroup each; their raw partial tiles go to a
// scratch buffer [K][S][B][N] and a finishing pass adds them up IN RANGE ORDER (bitwise reproducible) and applies the
// epilogue of the unsplit kernel (bias + ReLU, or the ReLU mask of the producer layer).
// ---------------------------------------------------------------------------------------
struct EpiPartialStore {
    float* part; int M, N;
    __device__ __forceinline__ void operator()(const f32x16& acc, int m0, int n0) const {
        const int lane = threadIdx.x & 63;
        const int n = n0 + (lane & 31);
        if (n >= N) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mfma_row(e, lane);
            if (m < M) part[(size_t)m * N + n] = acc[e];
        }
    }
};

// k range [k0, k1) of split z out of S over a reduction of `len` in steps of BK
__device__ __forceinline__ void split_range(int len, int BK, int S, int z, int& k0, int& k1) {
    const int steps = (len + BK - 1) / BK, per = (steps + S - 1) / S;
    k0 = min(z * per, steps) * BK;
    k1 = min(min((z + 1) * per, steps) * BK, len);
}

template <int BK, class T>
__global__ __launch_bounds__(256) void linear_fwd_splitk_kernel(LinearArgs a, int S) {
    __shared__ GemmSmem<32, 128, 2, BK, T> sm;
    const int c = blockIdx.z, rt = blockIdx.y / S, z = blockIdx.y % S;
    const float* x = a.x + c * a.x_cs;
    const float* w = a.w + c * a.w_cs;
    LoadRowMajorK al{x, a.B, a.In, a.In, (bool)(a.vec & 1), a.sidx ? a.sidx + c * a.s_cs : nullptr};
    LoadRowMajorK bl{w, a.Out, a.In, a.In, (bool)(a.vec & 2)};
    EpiPartialStore epi{a.part + ((size_t)c * S + z) * a.B * a.Out, a.B, a.Out};
    int k0, k1;
    split_range(a.In, BK, S, z, k0, k1);
    gemm_block_tile<32, 128, 1, 4, LoadRowMajorK, LoadRowMajorK, EpiPartialStore, false, 2, BK, T>(
        sm, al, bl, epi, rt * 32, blockIdx.x * 128, k0, k1);
}

template <int BK, class T>
__global__ __launch_bounds__(256) void linear_dgrad_splitk_kernel(DgradArgs a, float* part, int S) {
    __shared__ GemmSmem<32, 128, 2, BK, T> sm;
    const int c = blockIdx.z, rt = blockIdx.y / S, z = blockIdx.y % S;
    LoadRowMajorK al{a.dy + c * a.dy_cs, a.B, a.Out, a.Out, (bool)a.vec};
    LoadColMajorK bl{a.w + c * a.w_cs, a.In, a.Out, a.In};
    EpiPartialStore epi{part + ((size_t)c * S + z) * a.B * a.In, a.B, a.In};
    int k0, k1;
    split_range(a.Out, BK, S, z, k0, k1);
    gemm_block_tile<32, 128, 1, 4, LoadRowMajorK, LoadColMajorK, EpiPartialStore, false, 2, BK, T>(
        sm, al, bl, epi, rt * 32, blockIdx.x * 128, k0, k1);
}

// y[c][m][n] = epilogue(sum_z part[c][z][m][n]):  bias != null: + bias[n], then ReLU if `relu` (forward);
// act != null: zero where act[c][m][n] <= 0 (data gradient). One thread per output element, z ascending.
__global__ __launch_bounds__(256) void linear_splitk_finish_kernel(const float* part, int S, float* y, long y_cs,
                                                                   const float* bias, long b_cs, const float* act,
                                                                   long act_cs, int M, int N, int relu) {
    const int c = blockIdx.y;
    const long mn = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < mn; i += (long)gridDim.x * 256) {
        const float* p = part + (size_t)c * S * mn + i;
        float v = 0.f;
        for (int z = 0; z < S; ++z) v += p[(size_t)z * mn];
        if (bias != nullptr) {
            v += bias[c * b_cs + i % N];
            if (relu) v = fmaxf(v, 0.f);
        }
        if (act != nullptr && !(act[c * act_cs + i] > 0.f)) v = 0.f;
        y[c * y_cs + i] = v;
    }
}

struct WgradArgs {
    const float* dy; long dy_cs;  // [K][B][Out]
    const float* x;  long x_cs;   // [K][B][In]
    float* dw;       long dw_cs;  // [K][Out][In]  (grad slab view)
    float* db;       long db_cs;  // [K][Out]
    int B, In, Out;
    const int* sidx; long s_cs;
};

// x^T with one extra all-ones row at index In: the bias gradient drops out of the same
// MFMA pass as column In of the product.
struct LoadXTOnes {
    static constexpr bool ROW_CONTIG = true;
    const float* x; int In, B;
    const int* idx = nullptr;      // optional gather of the batch rows
    // branch-free: out-of-range elements read a clamped (valid) address and are replaced afterwards, so the eight
    // (gathered: two-level) loads of a chunk are all in flight together instead of one guarded load after the other
    __device__ __forceinline__ void load8(int row, int k, float (&v)[8]) const {
        const int rowc = min(row, In - 1);
        int src[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kc = min(k + j, B - 1);
            src[j] = idx ? idx[kc] : kc;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[(size_t)src[j] * In + rowc];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k + j < B) ? (row < In ? v[j] : (row == In ? 1.f : 0.f)) : 0.f;
    }
};

struct EpiWgradStore {
    static constexpr int BATCH = 1;
    float* dw; float* db; int M, N;  // M = Out, N = In (column N = bias)
    __device__ __forceinline__ void store4(float4 g, int m, int n) const {
        if (m >= M || n > N) return;
        if (n + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<float4*>(dw + (size_t)m * N + n) = g;
        } else {
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j < N) dw[(size_t)m * N + n + j] = gv[j];
                else if (n + j == N) db[m] = gv[j];
            }
        }
    }
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
template <class T>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(WgradArgs a) {
    __shared__ GemmSmem<128, 128, 1, GEMM_BK, T> sm;
    const int c = blockIdx.z;
    LoadColMajorK al{a.dy + c * a.dy_cs, a.Out, a.B, a.Out};
    LoadXTOnes bl{a.x + c * a.x_cs, a.In, a.B, a.sidx ? a.sidx + c * a.s_cs : nullptr};
    EpiWgradStore epi{a.dw + c * a.dw_cs, a.db + c * a.db_cs, a.Out, a.In};
    gemm_block_tile<128, 128, 4, 1, LoadColMajorK, LoadXTOnes, EpiWgradStore, true, 1, GEMM_BK, T>(
        sm, al, bl, epi, blockIdx.y * 128, blockIdx.x * 128, 0, a.B);
}

// ---------------------------------------------------------------------------------------
// Weight gradient fused with the Adam update. For the big layers (MNIST-CNN l1: 6.4 M weights
// per client) the gradient tile never leaves the accumulators: the epilogue reads p/m/v,
// applies Adam and writes p/m/v back, so the 25.7 MB/client gradient is neither written nor
// re-read (the un-fused pair moves 9 words per weight, this kernel 6).
// ---------------------------------------------------------------------------------------
struct AdamHyper {
    float lr, beta1, beta2, eps, weight_decay, inv_bc1, inv_sqrt_bc2;
    const float* bc;   // optional device {inv_bc1, inv_sqrt_bc2}
    int fresh;
};

template <bool FRESH>
struct EpiWgradAdam {
    float* w; float* mw; float* vw;     // [M][N] weight, first and second moment
    float* b; float* mb; float* vb;     // [M] bias and its moments (column N of the product)
    int M, N;
    float step, beta1, beta2, eps, wd, isb2;
    // FRESH: first step of a new optimizer - the moments are zero and are not read. A compile-time switch: as a
    // run-time flag it changed the register allocation of the ordinary step (104 -> 80 VGPRs, i.e. six instead of
    // four workgroups per CU) and the convolution blocks of the side branch no longer found room beside it.
    __device__ __forceinline__ void upd(float* p, float* m, float* v, float g) const {
        const float pv = *p;
        g += wd * pv;
        const float mv = beta1 * (FRESH ? 0.f : *m) + (1.f - beta1) * g;
        const float vv = beta2 * (FRESH ? 0.f : *v) + (1.f - beta2) * g * g;
        *m = mv; *v = vv;
        *p = pv - step * (mv / (sqrtf(vv) * isb2 + eps));
    }
    // Rows are processed in batches (gemm_block_tile, N4 epilogue): the p / m / v loads of BATCH rows fly together.
    // weights and moments are touched exactly once per step: non-temporal both ways (measured on
    // MI355X, 100 clients x 3136x2048: 2.96 ms plain, 3.09 nt loads only, 2.88 nt loads + stores)
    static constexpr int BATCH = 4;
    struct Row { f32x4 p, m, v; };
    __device__ __forceinline__ bool vec(int m, int n) const { return m < M && n + 3 < N && (N & 3) == 0; }
    __device__ __forceinline__ void load4(Row& r, int m, int n) const {
        r.p = f32x4{0.f, 0.f, 0.f, 0.f}; r.m = r.p; r.v = r.p;
        if (vec(m, n)) {
            const size_t o = (size_t)m * N + n;
            r.p = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + o));
            if constexpr (!FRESH) {
                r.m = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mw + o));
                r.v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vw + o));
            }
        }
    }
    __device__ __forceinline__ void finish4(Row& r, float4 g, int m, int n) const {
        if (m >= M || n > N) return;
        if (vec(m, n)) {
            const size_t o = (size_t)m * N + n;
            const float* gg = &g.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gr = gg[j] + wd * r.p[j];
                r.m[j] = beta1 * r.m[j] + (1.f - beta1) * gr;
                r.v[j] = beta2 * r.v[j] + (1.f - beta2) * gr * gr;
                r.p[j] -= step * (r.m[j] / (sqrtf(r.v[j]) * isb2 + eps));
            }
            __builtin_nontemporal_store(r.p, reinterpret_cast<f32x4*>(w + o));
            __builtin_nontemporal_store(r.m, reinterpret_cast<f32x4*>(mw + o));
            __builtin_nontemporal_store(r.v, reinterpret_cast<f32x4*>(vw + o));
        } else {
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j < N) { const size_t o = (size_t)m * N + n + j; upd(w + o, mw + o, vw + o, gv[j]); }
                else if (n + j == N) upd(b + m, mb + m, vb + m, gv[j]);
            }
        }
    }
};

struct WgradAdamArgs {
    const float* dy; long dy_cs;
    const float* x;  long x_cs;
    float* w; float* mw; float* vw; long w_cs;   // slab views sharing one row stride
    float* b; float* mb; float* vb; long b_cs;
    int B, In, Out;
    AdamHyper h;
    const int* sidx; long s_cs;
    int hiprio;
};

// WR x WC waves: block tile (32 WR) x (128 WC). Wider tiles (WC = 4: 32 x 512) turn the weight/moment
// traffic of a workgroup into 2 KB row segments instead of 512 B ones (fewer DRAM page activations).
// Occupancy is pinned to 2 waves per SIMD (two workgroups per CU) on purpose. The kernel is HBM-bound and fastest
// with two (2.87 ms; three: 2.99 ms), and the register file is shared with the side branch of the step: with three
// resident workgroups (what its 154 unified VGPRs allow) the 512-thread convolution blocks never find 2 x 64 free
// VGPRs per SIMD, and the stage-2 weight gradient waits the whole update out (10 clients: 302 -> 341 ms per round
// after an unrelated loader change had shrunk the allocation from ~176 to 154 registers). amdgpu_waves_per_eu(2, 2)
// does not pad the allocation, so RegisterReserve keeps N extra VGPRs live from the first to the last instruction.
template <int N>
struct RegisterReserve {
    float r[N];
    __device__ __forceinline__ RegisterReserve() {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_mov_b32 %0, 0" : "=v"(r[i]));
    }
    __device__ __forceinline__ void release() const {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" ::"v"(r[i]));
    }
};
template <>
struct RegisterReserve<0> {
    __device__ __forceinline__ void release() const {}
};
constexpr int PIN_UPDATE_MAX_CLIENTS = 80;    // measured: pinned wins by 5-8 % at 20-50 clients, 2 % at 64, ties at 80, loses 1.6 % at 100
constexpr int UPDATE_KERNEL_RESERVE = 12;      // -> 188 VGPRs: two waves per SIMD, 2 x 64 left for a convolution block
                                               // (pinned by tests/test_host_sanitizers.py::test_fused_update_occupancy)

template <int WR, int WC, bool FRESH, int RESERVE, class T>
__global__ __launch_bounds__(256, 2)
void linear_wgrad_adam_kernel(WgradAdamArgs a) {
    // The few instructions of this HBM-bound kernel go first: beside matrix-pipe-bound convolution waves (fp32 two-stream
    // step) its 64 MFMAs per wave otherwise queue behind theirs, every workgroup lives longer and the weight /
    // moment stream - the critical path of the step - thins out (measured: 3.0 -> 4.8 ms per step).
    if (a.hiprio) __builtin_amdgcn_s_setprio(3);
    RegisterReserve<RESERVE> reserve;
    constexpr int BM = 32 * WR, BN = 128 * WC;
    __shared__ GemmSmem<BM, BN, 1, GEMM_BK, T> sm;
    const int c = blockIdx.z;
    LoadColMajorK al{a.dy + c * a.dy_cs, a.Out, a.B, a.Out};
    LoadXTOnes bl{a.x + c * a.x_cs, a.In, a.B, a.sidx ? a.sidx + c * a.s_cs : nullptr};
    const float ib1 = a.h.bc ? a.h.bc[0] : a.h.inv_bc1;
    const float isb2 = a.h.bc ? a.h.bc[1] : a.h.inv_sqrt_bc2;
    const long wo = c * a.w_cs, bo = c * a.b_cs;
    EpiWgradAdam<FRESH> epi{a.w + wo, a.mw + wo, a.vw + wo, a.b + bo, a.mb + bo, a.vb + bo, a.Out, a.In,
                            a.h.lr * ib1, a.h.beta1, a.h.beta2, a.h.eps, a.h.weight_decay, isb2};
    gemm_block_tile<BM, BN, WR, WC, LoadColMajorK, LoadXTOnes, EpiWgradAdam<FRESH>, true, 1, GEMM_BK, T>(
        sm, al, bl, epi, blockIdx.y * BM, blockIdx.x * BN, 0, a.B);
    reserve.release();
}

// ---------------------------------------------------------------------------------------
// Fused update that also produces the layer's DATA gradient (fp32): one pass over the weights less per step.
//
//   dX[b][n] = sum_m dY[b][m] * W[m][n]
//
// reads the same W the Adam epilogue is about to read. The epilogue holds W in accumulator layout - lane l,
// register e = W[row (e&3) + 8(e>>2) + 4(l>>5)][4 (l&31) + j] for the four column tiles j - and that IS a valid B
// operand of v_mfma_f32_32x32x2_f32 for the k pair (row, row + 4): the order inside a reduction is free as long as
// both operands agree, so the A operand is dY[b = l & 31][that same row], one dword from the dY tile already in LDS.
// The weights go from the load that Adam needs straight into the matrix cores; no data moves.
// To keep the reduction over m inside one workgroup a block owns a 128-column strip over ALL rows (16 row tiles
// for 2048 outputs; the X tile is staged once, dY tiles are double-buffered), the four waves' partial sums are
// added through LDS in a fixed order (bitwise reproducible), and dX leaves as 512-byte rows.
// Cost: 64 more MFMAs per wave and row tile (as many as the weight gradient itself) in a kernel whose matrix cores
// idle 91 % of the time; saved: the separate data-gradient kernel's pass over the weights (2.6 GB for 100 clients).
// ---------------------------------------------------------------------------------------
// The epilogue is a software pipeline: the p / m / v loads of a row do not depend on the gradient, so they are
// issued D rows (D x 3 float4 per lane) ahead of their use - across row tiles too, i.e. they fly while the next
// tile's MFMAs run. Before (3 loads, wait, update, 3 stores, per row: 16 full memory latencies per tile with 3 KB
// per wave in flight) the kernel was bound by bytes in flight per CU, not by HBM: its time grew 1:1 as CUs were
// masked away (tools/probe/cumask_probe.py: 3.08 ms on 256 CUs, 5.0 on 128).
// FULL: the strip has 128 valid columns and M % 128 == 0 - no predicates at all. Otherwise out-of-range lanes load
// from a clamped (valid) address and are masked at the stores; column N is the bias (In % 4 == 0 is required, so a
// float4 is either wholly inside or wholly outside the weight matrix).
template <bool FRESH, bool FULL, int D>
__device__ __forceinline__ void fused_update_strip(const WgradAdamArgs& a, const EpiWgradAdam<FRESH>& epi, float* dx,
                                                   float (&sa)[2][128 * (GEMM_BK + 4)], float (&sb)[128 * (GEMM_BK + 4)],
                                                   const LoadColMajorK& al, int n0) {
    constexpr int BK = GEMM_BK, LDK = BK + 4;
    static_assert(16 % D == 0, "ring slots must map statically across row tiles");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int M = a.Out, N = a.In;
    const int MT = (M + 127) / 128;
    const int n = n0 + 4 * r;
    const bool col_ok = FULL || n < N;

    TileStager<128, 256, LoadColMajorK, false, BK, float> sta;

    f32x16 accx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[j][e] = 0.f;

    // ring of D rows in flight
    f32x4 rp[D], rm[D], rv[D];
    auto issue = [&](int mt, int e, int slot) {
        const int m = mt * 128 + wave * 32 + mfma_row(e, lane);
        const size_t o = (FULL || (col_ok && m < M)) ? (size_t)m * N + n : 0;
        rp[slot] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(epi.w + o));
        if constexpr (!FRESH) {
            rm[slot] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(epi.mw + o));
            rv[slot] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(epi.vw + o));
        }
    };
#pragma unroll
    for (int e = 0; e < D; ++e) issue(0, e, e);

    int cur = 0;
    // one 128-row tile; LAST (compile time: the last tile is peeled, so the wait counts of the others never have to
    // cover a path without prefetches) = nothing follows: no dY staging, no loads for the next tile
    auto tile = [&](int mt, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int m0 = mt * 128;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const float* as = sa[cur] + (wave * 32 + r) * LDK + h * 4;
        const float* bs = sb + r * LDK + h * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(as + kk * 8);
            f32x4 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDK + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bfr[j][e], acc[j], 0, 0, 0);
        }
        if constexpr (!LAST) sta.load(al, m0 + 128, 0);

        // epilogue: Adam on the 32 x 128 piece of this wave, the weights feeding the data gradient on the way
        const float* dyt = sa[cur] + (wave * 32) * LDK + r;          // dY[b = r][row] = dyt[row * LDK]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int slot = e % D;
            const int mloc = mfma_row(e, lane);
            const int m = m0 + wave * 32 + mloc;
            const bool ok = FULL || (col_ok && m < M);
            const size_t o = (size_t)m * N + n;
            f32x4 pv = rp[slot];
            f32x4 mv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!FRESH) { mv = rm[slot]; vv = rv[slot]; }
            if (e + D < 16) issue(mt, e + D, slot);
            else if constexpr (!LAST) issue(mt + 1, e + D - 16, slot);
            if constexpr (!FULL) {
                if (!ok) pv = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float ae = dyt[mloc * LDK];
#pragma unroll
            for (int j = 0; j < 4; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, pv[j], accx[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gr = acc[j][e] + epi.wd * pv[j];
                mv[j] = epi.beta1 * mv[j] + (1.f - epi.beta1) * gr;
                vv[j] = epi.beta2 * vv[j] + (1.f - epi.beta2) * gr * gr;
                pv[j] -= epi.step * (mv[j] / (sqrtf(vv[j]) * epi.isb2 + epi.eps));
            }
            if (ok) {
                __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(epi.w + o));
                __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(epi.mw + o));
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(epi.vw + o));
            }
        }
        if constexpr (!FULL) {
            if (n == N) {                     // the ones row of X^T: bias gradient
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wave * 32 + mfma_row(e, lane);
                    if (m < M) epi.upd(epi.b + m, epi.mb + m, epi.vb + m, acc[0][e]);
                }
            }
        }
        if constexpr (!LAST) sta.store(sa[cur ^ 1]);
        __syncthreads();
        cur ^= 1;
    };
#pragma unroll 1
    for (int mt = 0; mt + 1 < MT; ++mt) tile(mt, std::false_type{});
    tile(MT - 1, std::true_type{});

    // dX strip = sum of the four waves' partials, in wave order
    float* red = sa[0];                     // [32 b][128 n]; the loop's last barrier freed both dY buffers
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* q = red + mfma_row(e, lane) * 128 + 4 * r + j;
                    *q = (w == 0 ? 0.f : *q) + accx[j][e];
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int b = i >> 5, q = i & 31;
        if (b < a.B && n0 + 4 * q + 3 < N)
            *reinterpret_cast<f32x4*>(dx + (size_t)b * N + n0 + 4 * q) = *reinterpret_cast<const f32x4*>(red + b * 128 + 4 * q);
    }
}

template <bool FRESH, int D>
__global__ __launch_bounds__(256, D >= 4 ? 1 : 2)
void linear_wgrad_adam_dgrad_kernel(WgradAdamArgs a, float* dx_all, long dx_cs) {
    constexpr int BK = GEMM_BK, LDK = BK + lds_pad<float>();
    static_assert(LDK == GEMM_BK + 4, "fp32 LDS rows carry 4 floats of padding");
    __shared__ __attribute__((aligned(16))) float sa[2][128 * LDK];
    __shared__ __attribute__((aligned(16))) float sb[128 * LDK];
    const int c = blockIdx.y, n0 = blockIdx.x * 128;
    LoadColMajorK al{a.dy + c * a.dy_cs, a.Out, a.B, a.Out};
    LoadXTOnes bl{a.x + c * a.x_cs, a.In, a.B, a.sidx ? a.sidx + c * a.s_cs : nullptr};
    const float ib1 = a.h.bc ? a.h.bc[0] : a.h.inv_bc1;
    const float isb2 = a.h.bc ? a.h.bc[1] : a.h.inv_sqrt_bc2;
    const long wo = c * a.w_cs, bo = c * a.b_cs;
    const EpiWgradAdam<FRESH> epi{a.w + wo, a.mw + wo, a.vw + wo, a.b + bo, a.mb + bo, a.vb + bo, a.Out, a.In,
                                  a.h.lr * ib1, a.h.beta1, a.h.beta2, a.h.eps, a.h.weight_decay, isb2};
    {
        TileStager<128, 256, LoadXTOnes, true, BK, float> stb;
        stb.load(bl, n0, 0);
        stb.store(sb);
        TileStager<128, 256, LoadColMajorK, false, BK, float> sta;
        sta.load(al, 0, 0);
        sta.store(sa[0]);
    }
    __syncthreads();
    float* dx = dx_all + c * dx_cs;
    if (n0 + 128 <= a.In && (a.Out & 127) == 0) fused_update_strip<FRESH, true, D>(a, epi, dx, sa, sb, al, n0);
    else fused_update_strip<FRESH, false, D>(a, epi, dx, sa, sb, al, n0);
}

// ---------------------------------------------------------------------------------------
// Weight gradient fused with the SGD-momentum update (the FEMNIST models): same idea, 4 words per weight
// instead of 7 (gradient write + read and the separate pass over p / buf go away).
// ---------------------------------------------------------------------------------------
template <bool FRESH>
struct EpiWgradSgdm {
    float* w; float* mw; float* b; float* mb;
    int M, N;
    float lr, momentum, wd;
    __device__ __forceinline__ void upd(float* p, float* m, float g) const {
        const float pv = *p;
        const float mv = momentum * (FRESH ? 0.f : *m) + (g + wd * pv);
        *m = mv;
        *p = pv - lr * mv;
    }
    static constexpr int BATCH = 8;
    struct Row { f32x4 p, m; };
    __device__ __forceinline__ bool vec(int m, int n) const { return m < M && n + 3 < N && (N & 3) == 0; }
    __device__ __forceinline__ void load4(Row& r, int m, int n) const {
        r.p = f32x4{0.f, 0.f, 0.f, 0.f}; r.m = r.p;
        if (vec(m, n)) {
            const size_t o = (size_t)m * N + n;
            r.p = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + o));
            if constexpr (!FRESH) r.m = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mw + o));
        }
    }
    __device__ __forceinline__ void finish4(Row& r, float4 g, int m, int n) const {
        if (m >= M || n > N) return;
        if (vec(m, n)) {
            const size_t o = (size_t)m * N + n;
            const float* gg = &g.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r.m[j] = momentum * r.m[j] + (gg[j] + wd * r.p[j]);
                r.p[j] -= lr * r.m[j];
            }
            __builtin_nontemporal_store(r.p, reinterpret_cast<f32x4*>(w + o));
            __builtin_nontemporal_store(r.m, reinterpret_cast<f32x4*>(mw + o));
        } else {
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j < N) { const size_t o = (size_t)m * N + n + j; upd(w + o, mw + o, gv[j]); }
                else if (n + j == N) upd(b + m, mb + m, gv[j]);
            }
        }
    }
};

struct WgradSgdmArgs {
    const float* dy; long dy_cs;
    const float* x;  long x_cs;
    float* w; float* mw; long w_cs;
    float* b; float* mb; long b_cs;
    int B, In, Out;
    float lr, momentum, weight_decay;
    int fresh;
    const int* sidx; long s_cs;
};

template <bool FRESH, class T>
__global__ __launch_bounds__(256, 2)
void linear_wgrad_sgdm_kernel(WgradSgdmArgs a) {
    __shared__ GemmSmem<128, 128, 1, GEMM_BK, T> sm;
    const int c = blockIdx.z;
    LoadColMajorK al{a.dy + c * a.dy_cs, a.Out, a.B, a.Out};
    LoadXTOnes bl{a.x + c * a.x_cs, a.In, a.B, a.sidx ? a.sidx + c * a.s_cs : nullptr};
    const long wo = c * a.w_cs, bo = c * a.b_cs;
    EpiWgradSgdm<FRESH> epi{a.w + wo, a.mw + wo, a.b + bo, a.mb + bo, a.Out, a.In, a.lr, a.momentum, a.weight_decay};
    gemm_block_tile<128, 128, 4, 1, LoadColMajorK, LoadXTOnes, EpiWgradSgdm<FRESH>, true, 1, GEMM_BK, T>(
        sm, al, bl, epi, blockIdx.y * 128, blockIdx.x * 128, 0, a.B);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

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

}  // namespace

P2PFL_API int p2pfl_linear_wgrad_adam(const float* dy, long dy_cs, const float* x, long x_cs, float* w,
                                      float* mw, float* vw, long w_cs, float* b, float* mb, float* vb,
                                      long b_cs, int K, int B, int In, int Out, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int t, const float* bc,
                                      int fresh, const int* sidx, long s_cs, int prec, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0 || t <= 0) return -1;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    WgradAdamArgs a{dy, dy_cs, x, x_cs, w, mw, vw, w_cs, b, mb, vb, b_cs, B, In, Out,
                    AdamHyper{lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                              (float)(1.0 / sqrt(bc2)), bc, fresh}, sidx, s_cs, 0};
    {
        const char* e = getenv("P2PFL_UPDATE_PRIO");
        a.hiprio = e ? atoi(e) != 0 : 1;
    }
    // 128 x 128 tiles, 2 workgroups per CU: measured best of {128x128, 64x256, 32x512} x {2, 3, 4 per CU}
    // (2.87 / 2.90 / 3.31 ms and 2.87 / 2.99 / 4.09 ms for 100 clients x 3136x2048 on MI355X)
    const dim3 grid(cdiv(In + 1, 128), cdiv(Out, 128), K);
    // few clients: the side branch of the step is as long as this kernel -> leave its blocks room (see RegisterReserve);
    // many clients: the side branch has slack and three resident workgroups stream a little faster
    const char* pin_env = getenv("P2PFL_PIN_UPDATE");
    const bool pin = pin_env ? atoi(pin_env) != 0 : K <= PIN_UPDATE_MAX_CLIENTS;
    if (prec != PREC_BF16) {
        if (fresh) hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, true, 0, float>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, false, 0, float>), grid, dim3(256), 0, stream, a);
    } else if (pin) {
        constexpr int R = UPDATE_KERNEL_RESERVE;
        if (fresh) hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, true, R, __bf16>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, false, R, __bf16>), grid, dim3(256), 0, stream, a);
    } else {
        if (fresh) hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, true, 0, __bf16>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_wgrad_adam_kernel<4, 1, false, 0, __bf16>), grid, dim3(256), 0, stream, a);
    }
    P2PFL_RETURN_LAST_ERROR();
}

// Does the fused update + data gradient take K clients of a B x In x Out layer (the occupancy rule below; alignment of
// the actual buffers is checked by the launch itself)? The engine sizes its client groups with this very rule.
P2PFL_API int p2pfl_linear_wgrad_adam_dgrad_accepts(int K, int B, int In, int Out) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0 || B > GEMM_BK || (In & 3) || K > 65535) return 0;
    const char* fe = getenv("P2PFL_FUSE_DGRAD_FORCE");
    if (fe && atoi(fe) != 0) return 1;
    return (long)cdiv(In + 1, 128) * K >= 3L * num_cus();
}

// Fused update + data gradient (fp32 policies; see linear_wgrad_adam_dgrad_kernel). Returns -4 when the shape is
// outside what the kernel covers (the caller then runs linear_dgrad + linear_wgrad_adam): B <= 32, In % 4 == 0,
// 16-byte aligned weight / dx rows.
P2PFL_API int p2pfl_linear_wgrad_adam_dgrad(const float* dy, long dy_cs, const float* x, long x_cs, float* w,
                                            float* mw, float* vw, long w_cs, float* b, float* mb, float* vb,
                                            long b_cs, float* dx, long dx_cs, int K, int B, int In, int Out, float lr,
                                            float beta1, float beta2, float eps, float weight_decay, int t,
                                            const float* bc, int fresh, const int* sidx, long s_cs, int prec,
                                            hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0 || t <= 0) return -1;
    if (prec == PREC_BF16 || B > GEMM_BK || (In & 3) || (w_cs & 3) || (dx_cs & 3) || ((uintptr_t)w & 15) ||
        ((uintptr_t)mw & 15) || ((uintptr_t)vw & 15) || ((uintptr_t)dx & 15) || K > 65535)
        return -4;
    // a block walks ALL row tiles of its column strip: with few clients there are too few strips to fill the chip.
    // Decline below three workgroups per CU unless asked (force: tests) - measured with the 8-row ring (one resident
    // workgroup per CU), MNIST-CNN l1, rounds with / without: 10 clients 359 / 344 ms, 20: 64.6 / 61.6, 30: 82.6 / 83.4,
    // 40: 108.2 / 112.1 (tools/probe/force_dgrad.sh)
    const char* fe = getenv("P2PFL_FUSE_DGRAD_FORCE");
    const bool force = fe && atoi(fe) != 0;
    if (!force && (long)cdiv(In + 1, 128) * K < 3L * num_cus()) return -4;
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    WgradAdamArgs a{dy, dy_cs, x, x_cs, w, mw, vw, w_cs, b, mb, vb, b_cs, B, In, Out,
                    AdamHyper{lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                              (float)(1.0 / sqrt(bc2)), bc, fresh}, sidx, s_cs, 0};
    const dim3 grid(cdiv(In + 1, 128), K);
    // rows in flight per lane (ring depth of the epilogue pipeline). Alone, 8 rows = 24 float4 per lane win - they need
    // the whole register file (467 registers: ONE workgroup per CU): 100 clients x 3136x2048 on MI355X 3.26 ms (no
    // pipeline) -> 3.08 (depth 2, two workgroups per CU, spills) -> 3.01 ms = 5.2 TB/s, which is what ANY in-place
    // 3-read / 3-write fp32 stream reaches on this part whatever its geometry (tools/probe/stream_geom.hip). Inside the
    // two-group round 4 rows (376 registers, still one workgroup per CU, 136 registers per SIMD left for a co-resident
    // wave of the other group's chain) are as fast or faster everywhere (tools/probe/depth_ab.sh: headline 246.7 -> 244.9
    // ms, cnn100-encrypted-churn 244.1 -> 241.2, fp32-mfma headline 290.1 -> 286.3, 200 / 1000 nodes on the card 484.0 /
    // 483.6 and 2379 / 2373): the default. P2PFL_UPDATE_DEPTH=2 / 8 / 16 select the other variants (experiments).
    static const int depth = [] { const char* e = getenv("P2PFL_UPDATE_DEPTH"); return e ? atoi(e) : 4; }();
#define P2PFL_LAUNCH_FUSED(D_)                                                                                          \
    do {                                                                                                                \
        if (fresh) hipLaunchKernelGGL((linear_wgrad_adam_dgrad_kernel<true, D_>), grid, dim3(256), 0, stream, a, dx, dx_cs); \
        else hipLaunchKernelGGL((linear_wgrad_adam_dgrad_kernel<false, D_>), grid, dim3(256), 0, stream, a, dx, dx_cs);   \
    } while (0)
    if (depth == 2) P2PFL_LAUNCH_FUSED(2);
    else if (depth == 8) P2PFL_LAUNCH_FUSED(8);
    else if (depth == 16) P2PFL_LAUNCH_FUSED(16);
    else P2PFL_LAUNCH_FUSED(4);
#undef P2PFL_LAUNCH_FUSED
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_linear_wgrad_sgdm(const float* dy, long dy_cs, const float* x, long x_cs, float* w,
                                      float* mw, long w_cs, float* b, float* mb, long b_cs, int K, int B,
                                      int In, int Out, float lr, float momentum, float weight_decay,
                                      int fresh, const int* sidx, long s_cs, int prec, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    WgradSgdmArgs a{dy, dy_cs, x, x_cs, w, mw, w_cs, b, mb, b_cs, B, In, Out, lr, momentum, weight_decay, fresh,
                    sidx, s_cs};
    const dim3 grid(cdiv(In + 1, 128), cdiv(Out, 128), K);
    if (prec != PREC_BF16) {
        if (fresh) hipLaunchKernelGGL((linear_wgrad_sgdm_kernel<true, float>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_wgrad_sgdm_kernel<false, float>), grid, dim3(256), 0, stream, a);
    } else {
        if (fresh) hipLaunchKernelGGL((linear_wgrad_sgdm_kernel<true, __bf16>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_wgrad_sgdm_kernel<false, __bf16>), grid, dim3(256), 0, stream, a);
    }
    P2PFL_RETURN_LAST_ERROR();
}

// blocks the skinny kernel splits the reduction of one (client, row tile) over: up to 8, while the grid stays
// below one block per CU and every block keeps >= 16 k-steps per wave round
static int skinny_splits(int K, int B, int In) {
    const long tiles = (long)K * cdiv(B, 32);
    int ks = 1;
    while (ks < 8 && tiles * ks * 2 <= num_cus() && (In + 15) / 16 / (ks * 2) >= 16) ks *= 2;
    return ks;
}

// Ranges a tiled product with `tiles` workgroups splits its reduction of `len` into (see linear_fwd_splitk_kernel):
// as many as keep the grid within three workgroups per CU, each range >= 16 k-steps of 32; 1 = do not split.
// fp32 policies only (the bf16 kernels of a sparse grid take 128-deep steps instead). P2PFL_SPLITK=0 disables.
static int tile_splits(long tiles, int len, int prec) {
    static const bool off = [] { const char* e = getenv("P2PFL_SPLITK"); return e && atoi(e) == 0; }();
    if (off || prec == PREC_BF16 || tiles <= 0 || tiles > num_cus() || len < 1024) return 1;
    long s = 3L * num_cus() / tiles;
    const long cap = len / (16 * 32);
    if (s > cap) s = cap;
    if (s > 8) s = 8;
    return s < 2 ? 1 : (int)s;
}

// floats of scratch p2pfl_linear_fwd wants for this problem (0: none)
P2PFL_API long p2pfl_linear_fwd_scratch(int K, int B, int In, int Out) {
    if (K <= 0 || B <= 0) return 0;
    if (!(Out <= 64 && In >= 512 && (long)K * cdiv(B, 32) <= 2L * num_cus())) {
        const int S = tile_splits((long)cdiv(Out, 128) * cdiv(B, 32) * K, In, PREC_F32);
        return S > 1 ? (long)K * S * B * Out : 0;
    }
    const int ks = skinny_splits(K, B, In);
    return ks > 1 ? (long)K * cdiv(B, 32) * ks * (Out <= 32 ? 1 : 2) * 1024 : 0;
}

// floats of scratch p2pfl_linear_dgrad wants for this problem (0: none)
P2PFL_API long p2pfl_linear_dgrad_scratch(int K, int B, int In, int Out) {
    if (K <= 0 || B <= 0) return 0;
    const int S = tile_splits((long)cdiv(In, 128) * cdiv(B, 32) * K, Out, PREC_F32);
    return S > 1 ? (long)K * S * B * In : 0;
}

P2PFL_API int p2pfl_linear_fwd(const float* x, long x_cs, const float* w, long w_cs, const float* b,
                               long b_cs, float* y, long y_cs, int K, int B, int In, int Out,
                               int relu, const int* sidx, long s_cs, float* scratch, long scratch_floats,
                               int prec, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    // float4 loads need In % 4 == 0 and 16-B aligned bases/strides; otherwise element loads
    const int xv = !((In & 3) || (x_cs & 3) || ((uintptr_t)x & 15));
    const int wv = !((In & 3) || (w_cs & 3) || ((uintptr_t)w & 15));
    LinearArgs a{x, x_cs, w, w_cs, b, b_cs, y, y_cs, B, In, Out, relu, xv | (wv << 1), sidx, s_cs, scratch};
    // (with >= 2 tiles per CU the tiled kernel hides the latency itself: 1000 clients x 3136 -> 62: 259 vs 300 us)
    if (Out <= 64 && In >= 512 && (long)K * cdiv(B, 32) <= 2L * num_cus()) {
        int ks = skinny_splits(K, B, In);
        const int nt = Out <= 32 ? 1 : 2;
        if (scratch == nullptr || scratch_floats < (long)K * cdiv(B, 32) * ks * nt * 1024) ks = 1;
        const dim3 sgrid(ks, cdiv(B, 32), K), rgrid(1, cdiv(B, 32), K);
        const bool f32 = prec != PREC_BF16;
        if (nt == 1 && f32) hipLaunchKernelGGL((linear_fwd_skinny_kernel<1, 4, true>), sgrid, dim3(1024), 0, stream, a);
        else if (nt == 1) hipLaunchKernelGGL((linear_fwd_skinny_kernel<1, 4, false>), sgrid, dim3(1024), 0, stream, a);
        else if (f32) hipLaunchKernelGGL((linear_fwd_skinny_kernel<2, 2, true>), sgrid, dim3(1024), 0, stream, a);
        else hipLaunchKernelGGL((linear_fwd_skinny_kernel<2, 2, false>), sgrid, dim3(1024), 0, stream, a);
        if (ks > 1) {
            if (nt == 1) hipLaunchKernelGGL(linear_skinny_reduce_kernel<1>, rgrid, dim3(1024), 0, stream, a, ks);
            else hipLaunchKernelGGL(linear_skinny_reduce_kernel<2>, rgrid, dim3(1024), 0, stream, a, ks);
        }
        P2PFL_RETURN_LAST_ERROR();
    }
    dim3 grid(cdiv(Out, 128), cdiv(B, 32), K);
    {   // few tiles, long reduction: split it (fp32 policies; needs the caller's scratch)
        const int S = tile_splits((long)grid.x * grid.y * grid.z, In, prec);
        if (S > 1 && scratch != nullptr && scratch_floats >= (long)K * S * B * Out && K <= 65535) {
            const dim3 sgrid(grid.x, grid.y * S, K);
            hipLaunchKernelGGL((linear_fwd_splitk_kernel<32, float>), sgrid, dim3(256), 0, stream, a, S);
            const int fx = (int)(((long)B * Out + 255) / 256);
            hipLaunchKernelGGL(linear_splitk_finish_kernel, dim3(fx < 64 ? fx : 64, K), dim3(256), 0, stream, scratch, S, y,
                               y_cs, b, b_cs, (const float*)nullptr, 0L, B, Out, relu);
            P2PFL_RETURN_LAST_ERROR();
        }
    }
    // long reductions (MNIST-CNN l1: In = 3136) take 64-deep k-steps: half the barriers / twice the bytes in
    // flight per step (100 clients on MI355X: 546 -> 516 us; 128-deep steps cost occupancy: 514-589 us)
    // few clients: at most one block per CU, so the bytes in flight per block bound the bandwidth -> 128-deep
    const bool sparse_grid = (long)grid.x * grid.y * grid.z <= num_cus() && getenv("P2PFL_NO_BK128") == nullptr;
    // (single-buffered 64- / 128-deep variants: 528 us against 516-528 double-buffered 64-deep; 128-deep x 2: 640;
    //  32 x 64 tiles with 2 waves, i.e. twice the blocks: 549 us with 64-deep, 609 us with 128-deep steps)
    if (prec != PREC_BF16) {
        // fp32 tiles are twice the bytes: 32-deep steps x 2 buffers = the LDS footprint (46 KB, 3 workgroups per
        // CU) of the 64-deep bf16 variant; 64-deep (87 KB, one workgroup per CU) only for grids below one block per CU
        if (In >= 1024 && sparse_grid) hipLaunchKernelGGL((linear_fwd_kernel<64, 2, float>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_fwd_kernel<32, 2, float>), grid, dim3(256), 0, stream, a);
    }
    else if (In >= 1024 && sparse_grid) hipLaunchKernelGGL((linear_fwd_kernel<128, 2, __bf16>), grid, dim3(256), 0, stream, a);
    else if (In >= 1024) hipLaunchKernelGGL((linear_fwd_kernel<64, 2, __bf16>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((linear_fwd_kernel<32, 2, __bf16>), grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_linear_dgrad(const float* dy, long dy_cs, const float* w, long w_cs,
                                 const float* act, long act_cs, float* dx, long dx_cs, int K, int B,
                                 int In, int Out, float* scratch, long scratch_floats, int prec, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    // dy rows of e.g. 10 classes are not 16-B aligned: element loads then
    const int vec = !((Out & 3) || (dy_cs & 3) || ((uintptr_t)dy & 15));
    DgradArgs a{dy, dy_cs, w, w_cs, act, act_cs, dx, dx_cs, B, In, Out, vec};
    dim3 grid(cdiv(In, 128), cdiv(B, 32), K);
    {   // few tiles, long reduction: split it (see linear_dgrad_splitk_kernel)
        const int S = tile_splits((long)grid.x * grid.y * grid.z, Out, prec);
        if (S > 1 && scratch != nullptr && scratch_floats >= (long)K * S * B * In && K <= 65535) {
            const dim3 sgrid(grid.x, grid.y * S, K);
            hipLaunchKernelGGL((linear_dgrad_splitk_kernel<32, float>), sgrid, dim3(256), 0, stream, a, scratch, S);
            const int fx = (int)(((long)B * In + 255) / 256);
            hipLaunchKernelGGL(linear_splitk_finish_kernel, dim3(fx < 64 ? fx : 64, K), dim3(256), 0, stream, scratch, S, dx,
                               dx_cs, (const float*)nullptr, 0L, act, act_cs, B, In, 0);
            P2PFL_RETURN_LAST_ERROR();
        }
    }
    const bool sparse_grid = (long)grid.x * grid.y * grid.z <= num_cus() && getenv("P2PFL_NO_BK128") == nullptr;
    if (prec != PREC_BF16) {
        if (Out >= 1024 && sparse_grid) hipLaunchKernelGGL((linear_dgrad_kernel<64, 2, float>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_dgrad_kernel<32, 2, float>), grid, dim3(256), 0, stream, a);
    }
    else if (Out >= 1024 && sparse_grid) hipLaunchKernelGGL((linear_dgrad_kernel<128, 2, __bf16>), grid, dim3(256), 0, stream, a);
    else if (Out >= 1024) hipLaunchKernelGGL((linear_dgrad_kernel<64, 2, __bf16>), grid, dim3(256), 0, stream, a);   // 498 -> 478 us
    else hipLaunchKernelGGL((linear_dgrad_kernel<32, 2, __bf16>), grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_linear_wgrad(const float* dy, long dy_cs, const float* x, long x_cs, float* dw,
                                 long dw_cs, float* db, long db_cs, int K, int B, int In, int Out,
                                 const int* sidx, long s_cs, int prec, hipStream_t stream) {
    if (K <= 0 || B <= 0 || In <= 0 || Out <= 0) return -1;
    WgradArgs a{dy, dy_cs, x, x_cs, dw, dw_cs, db, db_cs, B, In, Out, sidx, s_cs};
    dim3 grid(cdiv(In + 1, 128), cdiv(Out, 128), K);
    if (prec != PREC_BF16) hipLaunchKernelGGL(linear_wgrad_kernel<float>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(linear_wgrad_kernel<__bf16>, grid, dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}
